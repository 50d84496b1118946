/* Pure-C consumer of the drop-in boundary: compiles include/flowdec_hip.h as C99, resolves every entry point a C / cgo /
 * JNI host would bind with dlsym, and calls the host-only ones (no GPU needed).  Built and run by
 * tests/test_host_cpu.py::test_c_abi_from_plain_c. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "flowdec_hip.h"

#define NEED(sym)                                                        \
  do {                                                                   \
    if (!dlsym(h, #sym)) { printf("missing symbol %s\n", #sym); return 2; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 64;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
  NEED(fd_last_error); NEED(fd_version); NEED(fd_upfirdn2d); NEED(fd_fused_bias_act); NEED(fd_fir_resample);
  NEED(fd_conv2d); NEED(fd_conv_pack_weights); NEED(fd_gn_finalize); NEED(fd_stft_compress); NEED(fd_decompress_istft);
  NEED(fd_model_create); NEED(fd_model_set_param); NEED(fd_model_finalize); NEED(fd_ncsnpp_forward); NEED(fd_ode_solve);
  NEED(fd_enhance); NEED(fd_score_enhance); NEED(fd_regression_enhance); NEED(fd_stft_plan_create); NEED(fd_stft_plan_destroy);
  NEED(fd_calibrate_mfma); NEED(fd_enhance_ragged); NEED(fd_stft_compress_ragged); NEED(fd_decompress_istft_ragged);
  NEED(fd_enhance_normfac_offset); NEED(fd_model_set_normalize); NEED(fd_profile_read_fir); NEED(fd_conv_in);

  int (*version)(void) = (int (*)(void))dlsym(h, "fd_version");
  int (*num_frames)(int, int) = (int (*)(int, int))dlsym(h, "fd_num_frames");
  int (*padded)(int) = (int (*)(int))dlsym(h, "fd_padded_frames");
  const char* (*last_error)(void) = (const char* (*)(void))dlsym(h, "fd_last_error");
  int (*model_create)(const fd_model_config*, fd_model**) = (int (*)(const fd_model_config*, fd_model**))dlsym(h, "fd_model_create");
  int (*num_draws)(const fd_score_config*) = (int (*)(const fd_score_config*))dlsym(h, "fd_score_num_draws");
  if (version() < 100) return 3;
  if (num_frames(96000, 384) != 251 || padded(251) != 256) return 4;       /* frame bookkeeping is integer exact */
  fd_model_config bad; memset(&bad, 0, sizeof bad); bad.nf = 7;
  fd_model* m = 0;
  if (model_create(&bad, &m) != FD_EINVAL || !strstr(last_error(), "nf")) return 5;   /* error code + message, no exception */
  fd_score_config sc; memset(&sc, 0, sizeof sc); sc.N = 30; sc.predictor = FD_PREDICTOR_REVERSE_DIFFUSION; sc.corrector = FD_CORRECTOR_ALD;
  sc.corrector_steps = 1;
  if (num_draws(&sc) != 61) return 6;
  printf("c abi ok (version %d)\n", version());
  return 0;
}
