// model.hip -- host side of the whole-model entry points: NCSN++ structure (ncsnpp.py:102-251),
// parameter store + MFMA weight packing, workspace planner, the forward pass (ncsnpp.py:254-399), the
// fixed-step ODE loop (model.py:503-515; torchdyn fixed-step semantics restated) with hipGraph capture, and
// FlowModel.enhance (model.py:476-528).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "common.h"
#include "internal.h"

namespace {

enum ModKind { M_GFP, M_LINEAR, M_CONV_IN, M_RB, M_COMBINE, M_GN, M_CONV_HEAD };

struct Mod {
  ModKind kind;
  int idx;              // index in all_modules
  int cin = 0, cout = 0;
  int c0 = 0, c1 = 0;   // virtual-concat split of cin (RB)
  bool up = false, down = false, has_c2 = false;
  int level = 0;                 // resolution level the block's convolutions run at (0 = full resolution)
  bool wino0 = false, wino1 = false;   // Conv_0 / Conv_1 (+ folded Conv_2) packed for the Winograd kernel
  void* w0w = nullptr; void* w1w = nullptr;   // FD_WINOGRAD_AUTO: second (Winograd) packing next to the direct one in w0 / w1
  void* w0w4 = nullptr; void* w1w4 = nullptr; // FD_WINOGRAD_AUTO: third packing, for the F(4,3) kernel (conv_wino4.hip), where its shape rules allow
  void* w0w44 = nullptr; void* w1w44 = nullptr; // FD_F32 | FD_WINOGRAD_AUTO: packing for the 2-D F(4x4, 3x3) float32 kernel (conv_wino44f.hip)
  // device pointers (filled by finalize)
  void* w0 = nullptr; void* w1 = nullptr; void* w2 = nullptr;   // packed conv weights
  float *gn0_g = nullptr, *gn0_b = nullptr, *gn1_g = nullptr, *gn1_b = nullptr;
  float* b1 = nullptr;                     // Conv_1 bias (+ Conv_2 bias when the shortcut conv is folded in)
  float* bias0_eff = nullptr;              // [nt][cout] Conv_0 bias + Dense_0(silu(temb)), model-owned scratch
  float *w_f32 = nullptr, *b_f32 = nullptr;  // small f32 weights (conv_in, combine, head bias, gn affine)
};

struct ParamInfo {
  std::string name;
  std::vector<int> shape;
  long long numel() const { long long n = 1; for (int s : shape) n *= s; return n; }
};

// first-fit offset allocator over the caller's workspace; kernels run in stream order so a block may be
// reused as soon as it is released in program order
class Arena {
  struct Blk { size_t off, size; bool free; };
  std::vector<Blk> blks_;
  size_t end_ = 0, peak_ = 0;
 public:
  size_t alloc(size_t bytes) {
    bytes = fd_align(bytes ? bytes : 1);
    for (size_t i = 0; i < blks_.size(); ++i)
      if (blks_[i].free && blks_[i].size >= bytes) {
        const size_t rest = blks_[i].size - bytes;
        blks_[i].free = false; blks_[i].size = bytes;
        if (rest) blks_.insert(blks_.begin() + i + 1, Blk{blks_[i].off + bytes, rest, true});
        return blks_[i].off;
      }
    if (!blks_.empty() && blks_.back().free) {  // grow the trailing free block
      blks_.back().free = false; blks_.back().size = bytes;
      end_ = blks_.back().off + bytes;
    } else {
      blks_.push_back(Blk{end_, bytes, false});
      end_ += bytes;
    }
    if (end_ > peak_) peak_ = end_;
    return blks_.back().off;
  }
  void release(size_t off) {
    for (size_t i = 0; i < blks_.size(); ++i)
      if (blks_[i].off == off && !blks_[i].free) {
        blks_[i].free = true;
        if (i + 1 < blks_.size() && blks_[i + 1].free) { blks_[i].size += blks_[i + 1].size; blks_.erase(blks_.begin() + i + 1); }
        if (i > 0 && blks_[i - 1].free) { blks_[i - 1].size += blks_[i].size; blks_.erase(blks_.begin() + i); }
        if (!blks_.empty() && blks_.back().free) { end_ = blks_.back().off; blks_.pop_back(); }
        return;
      }
  }
  size_t peak() const { return peak_; }
};

struct Tens {
  size_t off = (size_t)-1;
  int C = 0, H = 0, W = 0;
  size_t sums = (size_t)-1;  // offset of the per-tile partial (sum, sumsq) floats [B][tiles][stride][2], if computed
  int tiles = 0, stride = 0;
};

// Everything a captured solve bakes into its kernel arguments.  Compared field by field (a memcmp over the struct would read
// its padding bytes).
struct GraphKey {
  const void *Y = nullptr, *noise = nullptr, *X = nullptr, *traj = nullptr, *ws = nullptr, *y = nullptr, *xhat = nullptr, *lens = nullptr;
  int B = 0, T = 0, N = 0, solver = 0, L = 0, kind = 0, normalize = 1;
  float sigma_fac = 0.f;
  fd_score_config score{};   // kind 3 only (zero otherwise)
  auto tie() const {
    return std::tie(Y, noise, X, traj, ws, y, xhat, lens, B, T, N, solver, L, kind, normalize, sigma_fac, score.theta, score.sigma_min, score.sigma_max,
                    score.t_eps, score.snr, score.N, score.predictor, score.corrector, score.corrector_steps, score.denoise);
  }
  bool operator<(const GraphKey& o) const { return tie() < o.tie(); }
};

}  // namespace

struct fd_model {
  fd_model_config cfg;
  int dt = FD_BF16;                 // storage type (cfg.act_dtype without the algorithm flags)
  int n_freq = 0, temb_dim = 0;
  std::vector<Mod> mods;
  std::vector<ParamInfo> params;
  std::map<std::string, std::vector<float>> host;   // staged parameters
  std::map<std::string, float*> dev_f32;            // uploaded f32 copies
  std::vector<void*> dev_allocs;
  std::vector<double> sigma_host;
  double* sigma_dev = nullptr;
  int sigma_n = 0;
  bool finalized = false;
  float* temb = nullptr;            // [MAX_NT][temb_dim]
  fd_temb_job* jobs_dev = nullptr;
  int njobs = 0;
  float* wo = nullptr;              // output_layer weight [2][4]
  fd_stft_plan* stft = nullptr;
  std::map<GraphKey, hipGraphExec_t> graphs;
  std::set<GraphKey> seen;          // keys that ran once eagerly: a solve is captured at its SECOND sighting
  int normalize = 1;                // front end: 1 = per-clip max-abs normalisation ('noisy'), 0 = 'none'
  // second stream for the side branches of one network evaluation (time embedding, pyramid-head chain): forked from / joined into
  // the caller's stream with events, also inside a graph capture (parallel branches of the captured graph)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // profiling of the dominant kernel (conv MFMA)
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t ev_used = 0;
  double prof_flops = 0.0;
  double prof_flops_exec = 0.0;   // multiply-adds the chosen algorithm executes (Winograd launches: fewer than the direct count)
  double prof_bytes = 0.0;   // algorithmic HBM bytes of the timed launches: every operand read once + output written once
  // second class: the FIR resampling launches (HBM-bound): events + algorithmic bytes
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_fir;
  size_t ev_fir_used = 0;
  double prof_fir_bytes = 0.0;
  static constexpr int MAX_NT = 256;
  // One enqueueing call at a time per model: the entry points below share the model's scratch (time-embedding biases, graph cache,
  // side stream, profiling events).  A second thread entering while one is inside gets FD_EBUSY instead of corrupting that state;
  // callers that want concurrency create one fd_model per thread (weights are ~50 MB).  The reference's own native ops are
  // stateless and re-entrant (upfirdn2d_kernel.cu:224-231) -- so are fd_upfirdn2d / fd_conv2d / the other operator-level calls.
  std::atomic_flag busy = ATOMIC_FLAG_INIT;
};

namespace {
struct BusyGuard {
  fd_model* m; bool ok;
  explicit BusyGuard(fd_model* mm) : m(mm), ok(mm && !mm->busy.test_and_set(std::memory_order_acquire)) {}
  ~BusyGuard() { if (ok) m->busy.clear(std::memory_order_release); }
};
}  // namespace
#define FD_MODEL_ENTER(m, fn)                                                                                                   \
  FD_REQUIRE(m, fn ": null model");                                                                                             \
  BusyGuard fd_busy_guard_(m);                                                                                                  \
  if (!fd_busy_guard_.ok) return fd_set_error(FD_EBUSY, fn ": the model is serving a call of another thread (one enqueueing call at a time per fd_model)")

namespace {

int gn_groups(int C) { return (C / 4 < 32) ? C / 4 : 32; }

// mirrors NCSNpp.__init__ (ncsnpp.py:102-251) for progressive='output_skip', progressive_input='input_skip',
// combine 'sum', biggan blocks, no attention
void build_structure(fd_model* m) {
  const fd_model_config& c = m->cfg;
  const int nf = c.nf, R = c.num_levels, nrb = c.num_res_blocks, nch = 4;
  auto& mods = m->mods;
  auto& P = m->params;
  mods.clear(); P.clear();
  auto add_param = [&](const std::string& n, std::vector<int> s) { P.push_back(ParamInfo{n, s}); };
  auto pref = [](int i) { return "backbone.all_modules." + std::to_string(i) + "."; };
  add_param("backbone.output_layer.weight", {2, nch, 1, 1});
  int idx = 0;
  auto push = [&](Mod md) { md.idx = idx++; mods.push_back(md); return (int)mods.size() - 1; };
  auto add_rb = [&](int c0, int c1, int cout, bool up, bool down, int level) {
    Mod md{}; md.kind = M_RB; md.c0 = c0; md.c1 = c1; md.cin = c0 + c1; md.cout = cout; md.up = up; md.down = down; md.level = level;
    md.has_c2 = (md.cin != cout) || up || down;
    const int i = mods[push(md)].idx;
    add_param(pref(i) + "GroupNorm_0.weight", {md.cin}); add_param(pref(i) + "GroupNorm_0.bias", {md.cin});
    add_param(pref(i) + "Conv_0.weight", {cout, md.cin, 3, 3}); add_param(pref(i) + "Conv_0.bias", {cout});
    add_param(pref(i) + "Dense_0.weight", {cout, 4 * nf}); add_param(pref(i) + "Dense_0.bias", {cout});
    add_param(pref(i) + "GroupNorm_1.weight", {cout}); add_param(pref(i) + "GroupNorm_1.bias", {cout});
    add_param(pref(i) + "Conv_1.weight", {cout, cout, 3, 3}); add_param(pref(i) + "Conv_1.bias", {cout});
    if (md.has_c2) { add_param(pref(i) + "Conv_2.weight", {cout, md.cin, 1, 1}); add_param(pref(i) + "Conv_2.bias", {cout}); }
  };
  { Mod md{}; md.kind = M_GFP; push(md); add_param(pref(0) + "W", {nf}); }
  { Mod md{}; md.kind = M_LINEAR; md.cin = 2 * nf; md.cout = 4 * nf; push(md); add_param(pref(1) + "weight", {4 * nf, 2 * nf}); add_param(pref(1) + "bias", {4 * nf}); }
  { Mod md{}; md.kind = M_LINEAR; md.cin = 4 * nf; md.cout = 4 * nf; push(md); add_param(pref(2) + "weight", {4 * nf, 4 * nf}); add_param(pref(2) + "bias", {4 * nf}); }
  { Mod md{}; md.kind = M_CONV_IN; md.cin = nch; md.cout = nf; push(md); add_param(pref(3) + "weight", {nf, nch, 3, 3}); add_param(pref(3) + "bias", {nf}); }
  std::vector<int> hs_c{nf};
  int in_ch = nf;
  for (int lvl = 0; lvl < R; ++lvl) {
    for (int b = 0; b < nrb; ++b) {
      const int out_ch = nf * c.ch_mult[lvl];
      add_rb(in_ch, 0, out_ch, false, false, lvl);
      in_ch = out_ch;
      hs_c.push_back(in_ch);
    }
    if (lvl != R - 1) {
      add_rb(in_ch, 0, in_ch, false, true, lvl + 1);
      Mod md{}; md.kind = M_COMBINE; md.cin = nch; md.cout = in_ch;
      const int i = mods[push(md)].idx;
      add_param(pref(i) + "Conv_0.weight", {in_ch, nch, 1, 1}); add_param(pref(i) + "Conv_0.bias", {in_ch});
      hs_c.push_back(in_ch);
    }
  }
  in_ch = hs_c.back();
  add_rb(in_ch, 0, in_ch, false, false, R - 1);
  add_rb(in_ch, 0, in_ch, false, false, R - 1);
  for (int lvl = R - 1; lvl >= 0; --lvl) {
    for (int b = 0; b < nrb + 1; ++b) {
      const int out_ch = nf * c.ch_mult[lvl];
      const int sk = hs_c.back(); hs_c.pop_back();
      add_rb(in_ch, sk, out_ch, false, false, lvl);
      in_ch = out_ch;
    }
    { Mod md{}; md.kind = M_GN; md.cin = in_ch; const int i = mods[push(md)].idx;
      add_param(pref(i) + "weight", {in_ch}); add_param(pref(i) + "bias", {in_ch}); }
    { Mod md{}; md.kind = M_CONV_HEAD; md.cin = in_ch; md.cout = nch; const int i = mods[push(md)].idx;
      add_param(pref(i) + "weight", {nch, in_ch, 3, 3}); add_param(pref(i) + "bias", {nch}); }
    if (lvl != 0) add_rb(in_ch, 0, in_ch, true, false, lvl - 1);
  }
}

int upload_f32(fd_model* m, const std::string& name, float** out) {
  auto it = m->host.find(name);
  if (it == m->host.end()) return fd_set_error(FD_ESTATE, "parameter '%s' was never set", name.c_str());
  float* d = nullptr;
  FD_HIP(hipMalloc(&d, sizeof(float) * it->second.size()));
  FD_HIP(hipMemcpy(d, it->second.data(), sizeof(float) * it->second.size(), hipMemcpyHostToDevice));
  m->dev_allocs.push_back(d);
  m->dev_f32[name] = d;
  *out = d;
  return FD_OK;
}

int pack_conv(fd_model* m, const std::string& name, int Cout, int C0, int C1, int ks, const std::string& sc_name, int S0, int S1,
              void** out, hipStream_t st, int algo = 0) {
  float *src = nullptr, *sc = nullptr;
  FD_TRY(upload_f32(m, name, &src));
  if (!sc_name.empty()) FD_TRY(upload_f32(m, sc_name, &sc));
  void* dst = nullptr;
  algo |= m->cfg.act_dtype & (FD_BF16_OPERANDS | FD_BF16X3_OPERANDS);   // mixed / split modes: bf16 weights for f32 activations
  const long long bytes = fd_conv_packed_bytes(Cout, C0, C1, ks, S0, S1, m->dt | algo);
  FD_REQUIRE(bytes > 0, "internal: no packing for conv '%s'", name.c_str());
  FD_HIP(hipMalloc(&dst, (size_t)bytes));
  m->dev_allocs.push_back(dst);
  FD_TRY(fd_conv_pack_weights(src, sc, dst, Cout, C0, C1, ks, S0, S1, m->dt | algo, st));
  *out = dst;
  return FD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------------------------------------------
struct OutSpec {           // what to do with v = NCSNpp(x, y, t):  dst = base + coef * (v + kold); ksave = v
  const float* base = nullptr;
  const float* kold = nullptr;
  float coef = 1.f;
  float* dst = nullptr;
  float* ksave = nullptr;
  // score-sampler form (score = true):  dst = cb * base + cy * yv + coef * v + cz * z
  bool score = false;
  const float* yv = nullptr;
  const float* z = nullptr;
  float cb = 1.f, cy = 0.f, cz = 0.f;
};

struct Fwd {
  fd_model* m;
  bool dry;            // plan only (no launches, no pointers)
  char* base;
  Arena arena;
  hipStream_t st;
  int B, F, T, dt, esz;
  int w4_launches = 0;   // F(4,3) launches so far in this walk (alternating tile order)

  void* ptr(size_t off) const { return dry ? nullptr : base + off; }
  // ---- side branch: work that the main chain does not need right away runs on m->side between fork() and back(), and the main
  // stream waits for it at join().  Buffers a side branch frees go to `deferred` and are released at the join: until then the
  // arena may not hand them to a kernel of the main stream.  Off in profiling mode (per-kernel event timing) and for the planner.
  hipStream_t st_main = nullptr;
  std::vector<size_t> deferred;
  bool side_plan() const { return m && m->side; }                 // memory discipline: identical for the planner and every run
  bool side_ok() const { return !dry && side_plan() && !m->profiling; }   // the streams really fork
  int fork() {
    if (!side_ok()) return FD_OK;
    st_main = st;
    FD_HIP(hipEventRecord(m->ev_fork, st_main));
    FD_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    st = m->side;
    return FD_OK;
  }
  void back() { if (side_ok()) st = st_main; }
  int join() {
    if (side_ok()) {
      FD_HIP(hipEventRecord(m->ev_join, m->side));
      FD_HIP(hipStreamWaitEvent(st, m->ev_join, 0));
    }
    for (size_t off : deferred) arena.release(off);
    deferred.clear();
    return FD_OK;
  }
  void release_later(size_t off, bool on_side) { if (on_side) deferred.push_back(off); else arena.release(off); }
  Tens talloc(int C, int H, int W) { Tens t; t.C = C; t.H = H; t.W = W; t.off = arena.alloc((size_t)B * H * W * C * esz); return t; }
  void tfree(Tens& t) {
    if (t.off != (size_t)-1) arena.release(t.off);
    if (t.sums != (size_t)-1) arena.release(t.sums);
    t.off = t.sums = (size_t)-1;
  }
  int ensure_sums(Tens& t) {
    if (t.sums != (size_t)-1) return FD_OK;
    t.tiles = fd_channel_sums_tiles(t.H, t.W);
    t.stride = t.C;
    t.sums = arena.alloc(sizeof(float) * 2 * (size_t)B * t.tiles * t.stride);
    if (dry) return FD_OK;
    return fd_channel_sums(ptr(t.off), (float*)ptr(t.sums), B, t.H, t.W, t.C, dt, st);
  }
  // GroupNorm over the virtual concat [a | b] -> affine pairs
  int gn_affine(Tens& a, Tens* b, const float* gamma, const float* beta, size_t* aff_off) {
    FD_TRY(ensure_sums(a));
    if (b) FD_TRY(ensure_sums(*b));
    const int C = a.C + (b ? b->C : 0);
    *aff_off = arena.alloc(sizeof(float) * 2 * (size_t)B * C);
    if (dry) return FD_OK;
    return fd_gn_finalize((const float*)ptr(a.sums), a.tiles, a.stride, a.C, b ? (const float*)ptr(b->sums) : nullptr, b ? b->tiles : 0,
                          b ? b->stride : 0, b ? b->C : 0, gamma, beta, (float*)ptr(*aff_off), B, gn_groups(C), (long long)a.H * a.W, 1e-6f, st);
  }
  // out = scale * (conv_k(act([a|b])) + conv_1x1([s0|s1]) + bias + skip); optionally emits the GroupNorm partials of out
  // FD_WINOGRAD_AUTO: the Winograd kernel's 128-cout workgroups are half the size of the direct kernel's, so a small grid fills the
  // chip better with them (measured: 1.12-1.14x at 48 tiles per image x 8 clips, 0.93-0.97x at 192+ per image).  The choice looks at
  // the IMAGE only, never at the batch: a clip must give the same bits alone, in a batch, or in a shard of a batch (section 8(e)).
  bool auto_wino(const Tens& out) const { return fd_cdiv(out.H, 16) * fd_cdiv(out.W, 16) <= 96; }
  // FD_LOW_LATENCY (one short clip on the whole chip; profiles/r02_latency_tiles.txt): images of at most 24 tiles run the direct
  // kernel with 32-channel workgroups and chunk-resident weights (8 x the workgroups, one barrier per chunk); the F(2,3) kernel
  // (128-channel workgroups) takes everything else up to 128 tiles unless a 1x1 shortcut is folded in; those run the direct kernel with
  // 128-channel workgroups up to 128 tiles; above 128 tiles everything runs the F(4,3) kernel.  By image size only.
  int conv(const Tens& a, const Tens* b, size_t aff, const Tens* s0, const Tens* s1, const void* w, const float* bias, int bias_rows,
           const Tens* skip, float scale, Tens& out, int ks, bool want_stats, bool wino = false, const void* w_wino = nullptr,
           const void* w_wino4 = nullptr, const void* w_wino44 = nullptr) {
    int tile = 0;
    bool wino4 = false, wino44 = false;
    int order = 0;
    const int opflag = m ? (m->cfg.act_dtype & (FD_BF16_OPERANDS | FD_BF16X3_OPERANDS)) : 0;
    const int px_tiles = fd_cdiv(out.H, 16) * fd_cdiv(out.W, 16);
    const bool latency = m && (m->cfg.act_dtype & FD_LOW_LATENCY) && dt == FD_BF16;
    const bool autosel = m && (m->cfg.act_dtype & FD_WINOGRAD_AUTO) && dt == FD_BF16;
    // fp32 mode (pure f32: storage, operands, exact f32 MFMA): F(4,3) in float32 (conv_wino4f.hip) wherever its shape rules hold (Cout = 256,
    // whole 16 x 16 tiles, >= 64 input channels) -- the f32 matrix instruction is 16 x slower than the fp16 one, the launch is MFMA-bound at
    // any grid size, and both kernels use one 256-cout workgroup per tile: halving the MFMAs is worth 1.6-1.9 x per launch.  By shape only.
    const bool autosel_f32 = m && (m->cfg.act_dtype & FD_WINOGRAD_AUTO) && dt == FD_F32 && opflag == 0;
    // fp32 mode: 2-D Winograd F(4x4, 3x3) in float32 (conv_wino44f.hip: a quarter of the direct kernel's MFMAs) wherever its shape rules hold
    // (Cout % 128 == 0, whole 16 x 16 tiles, channel counts % 8 == 0): 1.1-1.4 x the F(4,3) float32 kernel per launch at 256 couts and
    // 1.8-2.2 x the direct kernel at 128 (profiles/r06_wino44f.txt); the matrix pipe bounds the launch at any grid size.  By shape only.
    if (m && (m->cfg.act_dtype & FD_WINOGRAD_AUTO) && dt == FD_F32 && opflag == 0 && w_wino44 && out.H % 16 == 0 && out.W % 16 == 0) { w = w_wino44; wino44 = true; }
    else if (latency && px_tiles <= 24 && out.C >= 64) tile = FD_TILE_BN32_CHUNK;
    else if (autosel && px_tiles <= 16 && out.C >= 64) tile = FD_TILE_BN64_CHUNK;   // the 96 x 32 level: 18.5 us vs 22.6 (Winograd) at 8 clips, 17.1 vs 21.9 at one
    // (never with a folded 1x1 shortcut: its input is the UN-NORMALISED residual stream, which the Winograd kernel would narrow to
    // the fp16 range; GroupNorm+SiLU outputs and their FIR-resampled versions are bounded)
    else if (w_wino && !s0 && (auto_wino(out) || (latency && px_tiles <= 128))) { w = w_wino; wino = true; }
    // images of more than 96 tiles (the two upper resolution levels of a 2 s clip): Winograd F(4,3) with 256-cout workgroups -- half the
    // MFMAs of the direct kernel, 1.09-1.19x per launch (profiles/r04_wino4_vs_direct.txt); whole 16 x 16 tiles only.  In `auto` mode this
    // branch ALSO takes the folded-shortcut Conv_1 launches of images of 17..96 tiles (the F(2,3) branch above excludes them): 1.17-1.20x
    // over the direct kernel at 192 x 64 x 8 clips (MEASUREMENTS R4.2, "L2" row).  One clip alone gives those launches 96 workgroups of
    // 512 threads on 256 CUs -- that case is what conv_algo = 'latency' (FD_TILE_BN128 below) is for; the choice must not look at B.
    // (the 64-channel input of the first block included: 1.07x with the halo of a chunk pair per request)
    // (a folded 1x1 shortcut runs as a bf16 GEMM on the raw residual stream in that kernel's epilogue: no fp16 range issue)
    // (FD_LOW_LATENCY: everything above 128 tiles: one 1 s clip 60.0 -> 63.4x, one 2 s clip 79 -> 88.8x real time)
    else if ((autosel || (latency && px_tiles > 128) || autosel_f32) && w_wino4 && !(s0 && skip) && out.H % 16 == 0 &&
             out.W % 16 == 0 && a.C + (b ? b->C : 0) >= 64) {
      w = w_wino4; wino4 = true;
      // every other F(4,3) launch of a forward walks its tiles backwards: a consumer then starts on the lines its producer wrote last,
      // which the memory-side cache still holds (the position in the launch sequence decides, so every forward has the same schedule)
      if (!dry && (w4_launches++ & 1) != 0) order = FD_TILE_REVERSED;   // (counted in launching walks only: a planning walk cannot shift the schedule)
    }
    // one clip, folded-shortcut convolutions of the 384 x 64 level (96 tiles): 96 workgroups of 256 channels leave 160 CUs idle; 128-channel
    // workgroups are 1.28-1.39x per launch there (scripts/ab_conv_b1.py with AB_H=384 AB_W=64), same bits
    else if (latency && px_tiles <= 128 && out.C >= 256 && out.C % 128 == 0) tile = FD_TILE_BN128;
    if (want_stats) {
      out.tiles = fd_conv_stats_tiles(out.H, out.W);
      out.stride = fd_conv_cout_pad(out.C);
      out.sums = arena.alloc(sizeof(float) * 2 * (size_t)B * out.tiles * out.stride);
    }
    if (dry) return FD_OK;
    if (m && m->profiling) {
      if (m->ev_used == m->ev.size()) {
        hipEvent_t e0, e1;
        FD_HIP(hipEventCreate(&e0)); FD_HIP(hipEventCreate(&e1));
        m->ev.emplace_back(e0, e1);
      }
      FD_HIP(hipEventRecord(m->ev[m->ev_used].first, st));
    }
    const int rc = fd_conv2d(ptr(a.off), a.C, b ? ptr(b->off) : nullptr, b ? b->C : 0, aff == (size_t)-1 ? nullptr : (const float*)ptr(aff),
                             s0 ? ptr(s0->off) : nullptr, s0 ? s0->C : 0, s1 ? ptr(s1->off) : nullptr, s1 ? s1->C : 0, w, bias, bias_rows,
                             skip ? ptr(skip->off) : nullptr, scale, ptr(out.off), out.C, want_stats ? (float*)ptr(out.sums) : nullptr, B,
                             out.H, out.W, ks, dt | (wino ? FD_WINOGRAD : 0) | (wino4 ? FD_WINOGRAD4 : 0) | (wino44 ? FD_WINOGRAD44 : 0) | order | tile | opflag, st);
    if (m && m->profiling) {
      FD_HIP(hipEventRecord(m->ev[m->ev_used].second, st));
      ++m->ev_used;
      m->prof_flops += 2.0 * B * out.H * out.W * (double)out.C *
                       ((a.C + (b ? b->C : 0)) * ks * ks + (s0 ? s0->C : 0) + (s1 ? s1->C : 0));
      const double cin = a.C + (b ? b->C : 0), csc = (s0 ? s0->C : 0) + (s1 ? s1->C : 0);
      // F(4,3): 6 products per 4 outputs and kernel row instead of 12 (the folded shortcut runs as a plain GEMM); F(2,3): 4 instead of 6
      // F(4x4, 3x3): 36 products per 16 outputs instead of 144
      m->prof_flops_exec += 2.0 * B * out.H * out.W * (double)out.C * (cin * ks * ks * (wino44 ? 0.25 : wino4 ? 0.5 : wino ? 2.0 / 3.0 : 1.0) + csc);
      m->prof_bytes += (double)B * out.H * out.W * esz * (cin + csc + (skip ? out.C : 0) + out.C) + (double)esz * out.C * (cin * ks * ks + csc);
    }
    return rc;
  }

  // fd_fir_resample with per-launch timing when profiling (bytes: input read once, every output written once)
  int fir(const void* x, const float* aff, void* o_raw, void* o_act, int H, int W, int C, int dir) {
    const bool prof = m && m->profiling;
    if (prof) {
      if (m->ev_fir_used == m->ev_fir.size()) {
        hipEvent_t e0, e1;
        FD_HIP(hipEventCreate(&e0)); FD_HIP(hipEventCreate(&e1));
        m->ev_fir.emplace_back(e0, e1);
      }
      FD_HIP(hipEventRecord(m->ev_fir[m->ev_fir_used].first, st));
    }
    const int rc = fd_fir_resample(x, aff, o_raw, o_act, B, H, W, C, dir, dt, st);
    if (prof) {
      FD_HIP(hipEventRecord(m->ev_fir[m->ev_fir_used].second, st));
      ++m->ev_fir_used;
      const double in = (double)B * H * W * C * esz, out1 = dir > 0 ? 4.0 * in : 0.25 * in;
      m->prof_fir_bytes += in + out1 * ((o_raw ? 1 : 0) + (o_act ? 1 : 0));
    }
    return rc;
  }

  int resblock(const Mod& md, Tens& x0, Tens* x1, int nt, Tens& out, bool out_given = false) {
    const float rs2 = 0.70710678118654752440f;
    size_t aff0;
    FD_TRY(gn_affine(x0, x1, md.gn0_g, md.gn0_b, &aff0));
    const int H = x0.H, W = x0.W;
    const int OH = md.up ? 2 * H : (md.down ? H / 2 : H), OW = md.up ? 2 * W : (md.down ? W / 2 : W);
    Tens h1 = talloc(md.cout, OH, OW);
    Tens xr, hr;
    if (md.up || md.down) {
      xr = talloc(md.cin, OH, OW); hr = talloc(md.cin, OH, OW);
      if (!dry) FD_TRY(fir(ptr(x0.off), (const float*)ptr(aff0), ptr(xr.off), ptr(hr.off), H, W, md.cin, md.up ? 1 : -1));
      FD_TRY(conv(hr, nullptr, (size_t)-1, nullptr, nullptr, md.w0, md.bias0_eff, nt, nullptr, 1.f, h1, 3, true, md.wino0, md.w0w, md.w0w4, md.w0w44));
      tfree(hr);
    } else {
      FD_TRY(conv(x0, x1, aff0, nullptr, nullptr, md.w0, md.bias0_eff, nt, nullptr, 1.f, h1, 3, true, md.wino0, md.w0w, md.w0w4, md.w0w44));
    }
    arena.release(aff0);
    size_t aff1;
    FD_TRY(gn_affine(h1, nullptr, md.gn1_g, md.gn1_b, &aff1));
    if (!out_given) out = talloc(md.cout, OH, OW);
    else { out.C = md.cout; out.H = OH; out.W = OW; out.sums = (size_t)-1; }   // fd_resblock: the caller's output tensor
    if (md.has_c2) {  // Conv_1(act(GN1(h))) + Conv_2(x) in one launch (shortcut conv folded in as extra K steps)
      if (md.up || md.down) FD_TRY(conv(h1, nullptr, aff1, &xr, nullptr, md.w1, md.b1, 1, nullptr, rs2, out, 3, true, md.wino1, md.w1w, md.w1w4, md.w1w44));
      else FD_TRY(conv(h1, nullptr, aff1, &x0, x1, md.w1, md.b1, 1, nullptr, rs2, out, 3, true, md.wino1, md.w1w, md.w1w4, md.w1w44));
    } else {
      FD_TRY(conv(h1, nullptr, aff1, nullptr, nullptr, md.w1, md.b1, 1, &x0, rs2, out, 3, true, md.wino1, md.w1w, md.w1w4, md.w1w44));
    }
    if (md.up || md.down) tfree(xr);
    arena.release(aff1);
    tfree(h1);
    return FD_OK;
  }

  // ncsnpp.py:254-399
  int run(const float* x, const float* y, const float* t, float t_imm, int nt, const OutSpec& os) {
    const fd_model_config& c = m->cfg;
    const int R = c.num_levels, nrb = c.num_res_blocks;
    const auto& mods = m->mods;
    const bool side = side_plan();
    if (!dry) {   // time embedding + the 22 Dense_0 biases: needed by the first ResBlock only -> beside the input packing / first conv
      if (side) FD_TRY(fork());
      FD_TRY(fd_time_embedding_impl(t, t_imm, nt, m->dev_f32["backbone.all_modules.0.W"], c.nf, m->dev_f32["backbone.all_modules.1.weight"],
                                    m->dev_f32["backbone.all_modules.1.bias"], m->dev_f32["backbone.all_modules.2.weight"],
                                    m->dev_f32["backbone.all_modules.2.bias"], m->temb, st));
      FD_TRY(fd_temb_bias_batched(m->jobs_dev, m->njobs, m->temb, nt, m->temb_dim, st));
      if (side) back();
    }
    size_t mi = 3;
    Tens in4 = talloc(8, F, T);   // 4 real channels + 4 zero channels (MFMA conv needs Cin % 8 == 0)
    if (!dry) { fd_edge_args a; a.x = x; a.y = y; a.out = ptr(in4.off); a.B = B; a.H = F; a.W = T; FD_TRY(fd_edge_op(0, a, dt, st)); }
    std::vector<Tens> hs;
    {
      const Mod& md = mods[mi++];
      Tens h0 = talloc(md.cout, F, T);
      // all_modules.3 (3x3, 4 -> nf), emitting the GroupNorm partials of its output.  bf16 mode, whole tiles: a vector-FMA kernel next to
      // its stores (conv_in_kernel: the layer is all prologue and epilogue on the matrix cores, 237 -> 60 us at 8 x 768 x 256); otherwise
      // the MFMA kernel on the zero-padded 8-channel input
      const bool vec_in = dt == FD_BF16 && md.w_f32 && F % 16 == 0 && T % 16 == 0 && (md.cout == 8 || md.cout == 16 || md.cout == 32 || md.cout == 64);
      if (vec_in) {
        h0.tiles = (F / 16) * (T / 16); h0.stride = md.cout;
        h0.sums = arena.alloc(sizeof(float) * 2 * (size_t)B * h0.tiles * h0.stride);
        if (!dry) { fd_edge_args a; a.x = ptr(in4.off); a.w = md.w_f32; a.bias = md.b_f32; a.out = ptr(h0.off); a.stats = (float*)ptr(h0.sums);
                    a.B = B; a.H = F; a.W = T; a.Cout = md.cout; FD_TRY(fd_edge_op(5, a, dt, st)); }
      } else {
        FD_TRY(conv(in4, nullptr, (size_t)-1, nullptr, nullptr, md.w0, md.b_f32, 1, nullptr, 1.f, h0, 3, true));
        if (!dry && m->profiling) {   // the 4 padding channels are not algorithmic work
          m->prof_flops -= 2.0 * B * F * T * (double)md.cout * 4 * 9;
          m->prof_flops_exec -= 2.0 * B * F * T * (double)md.cout * 4 * 9;
        }
      }
      hs.push_back(h0);
    }
    if (side) FD_TRY(join());   // the time-embedding biases are ready
    Tens pyr_in = in4;  // input pyramid (owned here)
    Tens h;
    for (int lvl = 0; lvl < R; ++lvl) {
      for (int b = 0; b < nrb; ++b) {
        FD_TRY(resblock(mods[mi++], hs.back(), nullptr, nt, h));
        hs.push_back(h);
      }
      if (lvl != R - 1) {
        Tens hd;
        FD_TRY(resblock(mods[mi++], hs.back(), nullptr, nt, hd));
        Tens p2 = talloc(8, pyr_in.H / 2, pyr_in.W / 2);
        if (!dry) FD_TRY(fir(ptr(pyr_in.off), nullptr, ptr(p2.off), nullptr, pyr_in.H, pyr_in.W, 8, -1));
        tfree(pyr_in);
        pyr_in = p2;
        const Mod& md = mods[mi++];
        Tens hc = talloc(md.cout, hd.H, hd.W);
        hc.tiles = fd_combine_tiles(hd.H, hd.W); hc.stride = md.cout;     // the combine kernel emits the GroupNorm partials of hc
        hc.sums = arena.alloc(sizeof(float) * 2 * (size_t)B * hc.tiles * hc.stride);
        if (!dry) { fd_edge_args a; a.x = ptr(pyr_in.off); a.y = ptr(hd.off); a.w = md.w_f32; a.bias = md.b_f32; a.out = ptr(hc.off);
                    a.stats = (float*)ptr(hc.sums); a.B = B; a.H = hd.H; a.W = hd.W; a.Cout = md.cout; FD_TRY(fd_edge_op(2, a, dt, st)); }
        tfree(hd);
        hs.push_back(hc);
      }
    }
    tfree(pyr_in);
    {
      Tens a1, a2;
      FD_TRY(resblock(mods[mi++], hs.back(), nullptr, nt, a1));
      FD_TRY(resblock(mods[mi++], a1, nullptr, nt, a2));
      tfree(a1);
      h = a2;
    }
    Tens pyramid; bool have_pyr = false;
    for (int lvl = R - 1; lvl >= 0; --lvl) {
      for (int b = 0; b < nrb + 1; ++b) {
        Tens sk = hs.back(); hs.pop_back();
        Tens o;
        FD_TRY(resblock(mods[mi++], h, &sk, nt, o));
        tfree(h); tfree(sk);
        h = o;
      }
      const Mod& gn = mods[mi++];
      const Mod& head = mods[mi++];
      // The head chain (GroupNorm + 3x3 conv to 4 channels + FIR-up of the running pyramid) only meets the main chain again at the
      // output layer: it runs on the side stream next to the up-sampling ResBlock that reads the same h.
      if (side) FD_TRY(fork());
      size_t aff;
      FD_TRY(gn_affine(h, nullptr, gn.gn0_g, gn.gn0_b, &aff));
      Tens pnew = talloc(4, h.H, h.W);
      if (have_pyr) {
        Tens pu = talloc(4, h.H, h.W);
        if (!dry) FD_TRY(fir(ptr(pyramid.off), nullptr, ptr(pu.off), nullptr, pyramid.H, pyramid.W, 4, +1));
        FD_TRY(conv(h, nullptr, aff, nullptr, nullptr, head.w0, head.b_f32, 1, &pu, 1.f, pnew, 3, false));
        release_later(pu.off, side); release_later(pyramid.off, side);
        pu.off = pyramid.off = (size_t)-1;
      } else {
        FD_TRY(conv(h, nullptr, aff, nullptr, nullptr, head.w0, head.b_f32, 1, nullptr, 1.f, pnew, 3, false));
      }
      release_later(aff, side);
      pyramid = pnew; have_pyr = true;
      if (side) back();
      if (lvl != 0) {
        Tens o;
        FD_TRY(resblock(mods[mi++], h, nullptr, nt, o));
        if (side) FD_TRY(join());   // the head has read h; its scratch goes back to the arena
        tfree(h);
        h = o;
      } else if (side) {
        FD_TRY(join());
      }
    }
    tfree(h);
    if (!hs.empty() || mi != mods.size()) return fd_set_error(FD_ESTATE, "internal: module walk mismatch");
    if (!dry) {
      fd_edge_args a; a.x = ptr(pyramid.off); a.w = m->wo; a.base = os.base; a.kold = os.kold; a.coef = os.coef; a.out = os.dst; a.ksave = os.ksave;
      a.B = B; a.H = F; a.W = T;
      if (os.score) { a.y = os.yv; a.z = os.z; a.cb = os.cb; a.cy = os.cy; a.cz = os.cz; }
      FD_TRY(fd_edge_op(os.score ? 4 : 3, a, dt, st));
    }
    tfree(pyramid);
    return FD_OK;
  }
};

int check_ready(const fd_model* m) {
  if (!m) return fd_set_error(FD_EINVAL, "null model");
  if (!m->finalized) return fd_set_error(FD_ESTATE, "model not finalised (call fd_model_finalize)");
  return FD_OK;
}

size_t forward_ws_bytes(const fd_model* m, int B, int T) {
  Fwd f{const_cast<fd_model*>(m), true, nullptr, Arena(), nullptr, B, m->n_freq, T, m->dt, (int)fd_dtype_size(m->dt)};
  OutSpec os;
  if (f.run(nullptr, nullptr, nullptr, 0.f, 1, os) != FD_OK) return 0;
  return f.arena.peak() + 256;
}

int check_shape(const fd_model* m, int B, int T) {
  const int div = 1 << (m->cfg.num_levels - 1);
  FD_REQUIRE(B > 0 && T > 0 && T % div == 0, "T_pad=%d must be a positive multiple of %d (B=%d)", T, div, B);
  FD_REQUIRE(m->n_freq % div == 0, "n_freq=%d not divisible by %d", m->n_freq, div);
  return FD_OK;
}

int forward_call(fd_model* m, const float* x, const float* y, const float* t, float t_imm, int nt, const OutSpec& os, int B, int T, void* ws,
                 size_t ws_bytes, hipStream_t st) {
  Fwd f{m, false, (char*)ws, Arena(), st, B, m->n_freq, T, m->dt, (int)fd_dtype_size(m->dt)};
  (void)ws_bytes;
  return f.run(x, y, t, t_imm, nt, os);
}

// torch.linspace(0, 1, N+1) in float32 with ATen's fused multiply-add evaluation (see oracle t_span_linspace)
std::vector<float> t_span_linspace(int N) {
  const int steps = N + 1;
  std::vector<float> out(steps);
  const float step = 1.0f / (float)(steps - 1);
  const int half = steps / 2;
  for (int i = 0; i < steps; ++i) out[i] = i < half ? fmaf(step, (float)i, 0.0f) : fmaf(-step, (float)(steps - 1 - i), 1.0f);
  return out;
}

int solver_nfe(int solver, int N) {
  switch (solver) {
    case FD_SOLVER_EULER: return N;
    case FD_SOLVER_MIDPOINT: case FD_SOLVER_HEUN2: return 2 * N;
    case FD_SOLVER_HEUN2_EULERLAST: return 2 * N - 1;
  }
  return -1;
}

size_t ode_ws_bytes(const fd_model* m, int B, int T) {
  const size_t state = fd_align(sizeof(float) * 2 * (size_t)B * m->n_freq * T);
  return 2 * state + forward_ws_bytes(m, B, T);
}

// the solver loop; everything is enqueued on `st` (eagerly or inside a capture)
int ode_enqueue(fd_model* m, const float* Y, const float* noise, float sigma_fac, int N, int solver, float* X, float* traj, int B, int T, void* ws,
                size_t ws_bytes, hipStream_t st) {
  const size_t nstate = (size_t)B * m->n_freq * T;
  const size_t state = fd_align(sizeof(float) * 2 * nstate);
  float* xtmp = (float*)ws;
  float* k1 = (float*)((char*)ws + state);
  void* fws = (char*)ws + 2 * state;
  const size_t fws_bytes = ws_bytes - 2 * state;
  FD_TRY(fd_init_state(Y, noise, m->sigma_dev, m->sigma_n, sigma_fac, X, B, m->n_freq, T, st));
  if (traj) FD_HIP(hipMemcpyAsync(traj, X, sizeof(float) * 2 * nstate, hipMemcpyDeviceToDevice, st));
  const std::vector<float> ts = t_span_linspace(N);
  float t = ts[0];
  float dt = ts[1] - ts[0];
  for (int i = 1; i <= N; ++i) {
    OutSpec os;
    if (solver == FD_SOLVER_EULER) {
      os.base = X; os.coef = dt; os.dst = X;
      FD_TRY(forward_call(m, X, Y, nullptr, t, 1, os, B, T, fws, fws_bytes, st));
    } else if (solver == FD_SOLVER_MIDPOINT) {
      const float half = 0.5f * dt;
      os.base = X; os.coef = half; os.dst = xtmp;
      FD_TRY(forward_call(m, X, Y, nullptr, t, 1, os, B, T, fws, fws_bytes, st));
      OutSpec o2; o2.base = X; o2.coef = dt; o2.dst = X;
      FD_TRY(forward_call(m, xtmp, Y, nullptr, t + half, 1, o2, B, T, fws, fws_bytes, st));
    } else {  // heun2 / heun2_eulerlast (sampling/solvers.py:15-57)
      const bool last = solver == FD_SOLVER_HEUN2_EULERLAST && fabsf((t + dt) - 1.0f) <= 1e-8f + 1e-5f;
      if (last) {
        os.base = X; os.coef = dt; os.dst = X;
        FD_TRY(forward_call(m, X, Y, nullptr, t, 1, os, B, T, fws, fws_bytes, st));
      } else {
        os.base = X; os.coef = dt; os.dst = xtmp; os.ksave = k1;
        FD_TRY(forward_call(m, X, Y, nullptr, t, 1, os, B, T, fws, fws_bytes, st));
        OutSpec o2; o2.base = X; o2.kold = k1; o2.coef = dt * 0.5f; o2.dst = X;
        FD_TRY(forward_call(m, xtmp, Y, nullptr, t + dt, 1, o2, B, T, fws, fws_bytes, st));
      }
    }
    if (traj) FD_HIP(hipMemcpyAsync(traj + 2 * nstate * i, X, sizeof(float) * 2 * nstate, hipMemcpyDeviceToDevice, st));
    t = t + dt;
    if (i < N) dt = ts[i + 1] - t;
  }
  return FD_OK;
}

// ---- ScoreDec baseline: OUVE closed forms + predictor-corrector sampler -------------------
// float32 scalar arithmetic like the reference's [B]-shaped tensors (sdes.py:168-192)
float ouve_std(const fd_score_config& c, float t) {
  const float th = c.theta, ls = logf(c.sigma_max / c.sigma_min);
  const float num = c.sigma_min * c.sigma_min * expf(-2.f * th * t) * (expf(2.f * (th + ls) * t) - 1.f) * ls;
  return sqrtf(num / (th + ls));
}
float ouve_diffusion(const fd_score_config& c, float t) {
  const float ls = logf(c.sigma_max / c.sigma_min);
  return c.sigma_min * powf(c.sigma_max / c.sigma_min, t) * sqrtf(2.f * ls);
}
// torch.linspace(start, end, steps) in float32 (see oracle linspace_f32)
std::vector<float> linspace_f32(float start, float end, int steps) {
  std::vector<float> out(steps);
  if (steps == 1) { out[0] = start; return out; }
  const float step = (end - start) / (float)(steps - 1);
  const int half = steps / 2;
  for (int i = 0; i < steps; ++i) out[i] = i < half ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - 1 - i), end);
  return out;
}
int score_draws(const fd_score_config& c) {
  return 1 + c.N * ((c.corrector == FD_CORRECTOR_ALD ? c.corrector_steps : 0) + (c.predictor != FD_PREDICTOR_NONE ? 1 : 0));
}

// sampling/__init__.py:57-70.  noise = [draws][B][F][T] complex64, consumed in the reference's order of randn_like calls.
int score_enqueue(fd_model* m, const float* Y, const float* noise, const fd_score_config& c, float* X, int B, int T, void* ws, size_t ws_bytes,
                  hipStream_t st) {
  const size_t nstate = (size_t)B * m->n_freq * T;
  const float* z = noise;
  auto next_z = [&]() { const float* r = z; z += 2 * nstate; return r; };
  FD_TRY(fd_caxpy(Y, next_z(), ouve_std(c, 1.0f), X, (long long)nstate, st));                 // prior, sdes.py:197-202
  const std::vector<float> ts = linspace_f32(1.0f, c.t_eps, c.N);
  for (int i = 0; i < c.N; ++i) {
    const float t = ts[i], std_t = ouve_std(c, t);
    if (c.corrector == FD_CORRECTOR_ALD) {                                                      // correctors.py:52-66
      for (int k = 0; k < c.corrector_steps; ++k) {
        const float step = (c.snr * std_t) * (c.snr * std_t) * 2.f;
        OutSpec os; os.score = true; os.base = X; os.dst = X; os.coef = -step / std_t; os.z = next_z(); os.cz = sqrtf(step * 2.f);
        FD_TRY(forward_call(m, X, Y, nullptr, t, 1, os, B, T, ws, ws_bytes, st));
      }
    }
    const bool last = i == c.N - 1 && c.denoise;                                                // result = x_mean of the last predictor step
    if (c.predictor == FD_PREDICTOR_REVERSE_DIFFUSION) {                                        // predictors.py:61-71, sdes.py:62-77,111-116
      const float dt = 1.f / (float)c.N, G = ouve_diffusion(c, t) * sqrtf(dt);
      OutSpec os; os.score = true; os.base = X; os.dst = X; os.yv = Y;
      os.cb = 1.f + c.theta * dt; os.cy = -c.theta * dt; os.coef = -(G * G) / std_t; os.z = next_z(); os.cz = last ? 0.f : G;
      FD_TRY(forward_call(m, X, Y, nullptr, t, 1, os, B, T, ws, ws_bytes, st));
    } else if (c.predictor == FD_PREDICTOR_EULER_MARUYAMA) {                                    // predictors.py:48-58, sdes.py:93-109
      const float dt = -1.f / (float)c.N, g = ouve_diffusion(c, t);
      OutSpec os; os.score = true; os.base = X; os.dst = X; os.yv = Y;
      os.cb = 1.f - c.theta * dt; os.cy = c.theta * dt; os.coef = g * g * dt / std_t; os.z = next_z(); os.cz = last ? 0.f : g * sqrtf(-dt);
      FD_TRY(forward_call(m, X, Y, nullptr, t, 1, os, B, T, ws, ws_bytes, st));
    }
  }
  return FD_OK;
}

template <typename Fn>
int run_maybe_graph(fd_model* m, const GraphKey& key, bool use_graph, hipStream_t st, Fn&& enqueue) {
  if (!use_graph || m->profiling) return enqueue();
  auto it = m->graphs.find(key);
  if (it == m->graphs.end()) {
    // A caller that walks a data set (one clip length per file) would capture and instantiate a ~1600-node graph per call and
    // never replay it: the first sighting of a key runs eagerly, only a repeated key is captured.
    if (!m->seen.count(key)) {
      if (m->seen.size() >= 256) m->seen.clear();
      m->seen.insert(key);
      return enqueue();
    }
    if (m->graphs.size() >= 32) {   // the key holds raw buffer pointers: bound the cache for callers that keep changing them
      FD_HIP(hipStreamSynchronize(st));   // (rare path) nothing captured earlier may still be in flight when it is destroyed
      for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second);
      m->graphs.clear();
    }
    hipGraph_t graph = nullptr;
    FD_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue();
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != FD_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fd_set_error(FD_ERUNTIME, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    FD_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    it = m->graphs.emplace(key, exec).first;
  }
  FD_HIP(hipGraphLaunch(it->second, st));
  return FD_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int fd_model_create(const fd_model_config* cfg, fd_model** out) {
  FD_REQUIRE(cfg && out, "fd_model_create: null pointer");
  FD_REQUIRE(cfg->nf >= 8 && cfg->nf % 8 == 0 && cfg->nf <= 64, "fd_model_create: nf must be a multiple of 8 in [8, 64] (got %d)", cfg->nf);
  FD_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= 8 && cfg->num_res_blocks >= 1, "fd_model_create: bad level / block counts");
  const int act_nos = cfg->act_dtype & ~FD_NO_SIDE_STREAM;
  FD_REQUIRE(((act_nos & 0xff) == FD_BF16 && !(act_nos & (FD_BF16_OPERANDS | FD_BF16X3_OPERANDS))) || act_nos == FD_F32 || act_nos == (FD_F32 | FD_WINOGRAD_AUTO) ||
                 act_nos == (FD_F32 | FD_BF16_OPERANDS) || act_nos == (FD_F32 | FD_BF16X3_OPERANDS),
             "fd_model_create: act_dtype must be FD_BF16 [| FD_WINOGRAD | FD_WINOGRAD_LOWRES | FD_WINOGRAD_AUTO | FD_LOW_LATENCY], FD_F32 [| FD_WINOGRAD_AUTO], "
             "FD_F32 | FD_BF16_OPERANDS or FD_F32 | FD_BF16X3_OPERANDS");
  {
    const int algo = cfg->act_dtype & (FD_WINOGRAD | FD_WINOGRAD_LOWRES | FD_WINOGRAD_AUTO | FD_LOW_LATENCY);
    FD_REQUIRE((algo & (algo - 1)) == 0, "fd_model_create: at most one of FD_WINOGRAD / FD_WINOGRAD_LOWRES / FD_WINOGRAD_AUTO / FD_LOW_LATENCY (got 0x%x)", algo);
    FD_REQUIRE(algo == 0 || (cfg->act_dtype & 0xff) == FD_BF16 || (algo == FD_WINOGRAD_AUTO && act_nos == (FD_F32 | FD_WINOGRAD_AUTO)),
               "fd_model_create: the convolution-algorithm flags go with FD_BF16 storage (FD_WINOGRAD_AUTO also with pure FD_F32: F(4,3) in float32)");
    FD_REQUIRE((cfg->act_dtype & FD_TILE_MASK) == 0, "fd_model_create: FD_TILE_* selects the workgroup width of ONE fd_conv2d launch, not of a model");
    FD_REQUIRE((cfg->act_dtype & ~(0xff | FD_WINOGRAD | FD_WINOGRAD_LOWRES | FD_WINOGRAD_AUTO | FD_LOW_LATENCY | FD_BF16_OPERANDS | FD_BF16X3_OPERANDS |
                                  FD_NO_SIDE_STREAM)) == 0, "fd_model_create: unknown bits in act_dtype (0x%x)", cfg->act_dtype);
  }
  FD_REQUIRE(cfg->n_fft > 0 && cfg->n_fft % 2 == 0 && cfg->hop > 0, "fd_model_create: bad STFT geometry");
  for (int i = 0; i < cfg->num_levels; ++i) {
    const int ch = cfg->nf * cfg->ch_mult[i];
    FD_REQUIRE(ch >= 8 && ch <= 256 && (ch & (ch - 1)) == 0, "fd_model_create: level width nf*ch_mult=%d must be a power of two in [8, 256]", ch);
  }
  fd_model* m = new fd_model();
  m->cfg = *cfg;
  m->dt = cfg->act_dtype & 0xff;
  m->n_freq = cfg->n_fft / 2 + 1;
  m->temb_dim = 4 * cfg->nf;
  build_structure(m);
  *out = m;
  return FD_OK;
}

extern "C" void fd_model_destroy(fd_model* m) {
  if (!m) return;
  for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second);
  for (void* p : m->dev_allocs) (void)hipFree(p);
  for (auto& e : m->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : m->ev_fir) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  fd_stft_plan_destroy(m->stft);
  if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
  if (m->ev_join) (void)hipEventDestroy(m->ev_join);
  if (m->side) (void)hipStreamDestroy(m->side);
  delete m;
}

extern "C" int fd_model_num_params(const fd_model* m) { return m ? (int)m->params.size() : 0; }

extern "C" int fd_model_param_info(const fd_model* m, int i, const char** name, int* ndim, int shape[4]) {
  FD_REQUIRE(m && i >= 0 && i < (int)m->params.size(), "fd_model_param_info: index out of range");
  const ParamInfo& p = m->params[i];
  if (name) *name = p.name.c_str();
  if (ndim) *ndim = (int)p.shape.size();
  if (shape) for (size_t k = 0; k < 4; ++k) shape[k] = k < p.shape.size() ? p.shape[k] : 1;
  return FD_OK;
}

extern "C" int fd_model_set_param(fd_model* m, const char* name, const float* host_data, long long numel) {
  FD_REQUIRE(m && name && host_data, "fd_model_set_param: null pointer");
  if (m->finalized) return fd_set_error(FD_ESTATE, "fd_model_set_param: model already finalised");
  for (const ParamInfo& p : m->params)
    if (p.name == name) {
      FD_REQUIRE(p.numel() == numel, "fd_model_set_param: '%s' expects %lld elements, got %lld", name, p.numel(), numel);
      m->host[name].assign(host_data, host_data + numel);
      return FD_OK;
    }
  return fd_set_error(FD_EINVAL, "fd_model_set_param: unknown parameter '%s'", name);
}

extern "C" int fd_model_set_sigma_y(fd_model* m, const double* host_sigma, int n) {
  FD_REQUIRE(m && host_sigma, "fd_model_set_sigma_y: null pointer");
  FD_REQUIRE(n == 1 || n == m->n_freq, "fd_model_set_sigma_y: expected 1 or %d values, got %d", m->n_freq, n);
  m->sigma_host.assign(host_sigma, host_sigma + n);
  m->sigma_n = n;
  if (m->sigma_dev) {
    FD_HIP(hipDeviceSynchronize());   // (rare, init-time) captured solves bake sigma_n into their launches: drop them
    for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second);
    m->graphs.clear(); m->seen.clear();
    FD_HIP(hipMemcpy(m->sigma_dev, m->sigma_host.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  }
  return FD_OK;
}

extern "C" int fd_model_set_normalize(fd_model* m, int normalize) {
  FD_REQUIRE(m, "fd_model_set_normalize: null model");
  m->normalize = normalize != 0;
  return FD_OK;
}

extern "C" int fd_model_finalize(fd_model* m, void* stream) {
  FD_REQUIRE(m, "fd_model_finalize: null model");
  if (m->finalized) return FD_OK;
  hipStream_t st = fd_stream(stream);
  for (const ParamInfo& p : m->params)
    if (!m->host.count(p.name)) return fd_set_error(FD_ESTATE, "fd_model_finalize: parameter '%s' missing", p.name.c_str());
  FD_TRY(fd_conv_init_attributes());
  auto pref = [](int i) { return "backbone.all_modules." + std::to_string(i) + "."; };
  float* tmp;
  for (int i = 0; i < 3; ++i) {
    if (i == 0) FD_TRY(upload_f32(m, pref(0) + "W", &tmp));
    else { FD_TRY(upload_f32(m, pref(i) + "weight", &tmp)); FD_TRY(upload_f32(m, pref(i) + "bias", &tmp)); }
  }
  FD_TRY(upload_f32(m, "backbone.output_layer.weight", &m->wo));
  std::vector<fd_temb_job> jobs;
  size_t bias_total = 0;
  for (Mod& md : m->mods) if (md.kind == M_RB) bias_total += (size_t)fd_model::MAX_NT * md.cout;
  float* bias_pool = nullptr;
  FD_HIP(hipMalloc(&bias_pool, sizeof(float) * (bias_total ? bias_total : 1)));
  m->dev_allocs.push_back(bias_pool);
  size_t bias_off = 0;
  for (Mod& md : m->mods) {
    const std::string p = pref(md.idx);
    switch (md.kind) {
      case M_CONV_IN: {
        // zero-pad the 4 input channels to 8 and pack for the MFMA conv
        const std::vector<float>& w4 = m->host[p + "weight"];
        std::vector<float>& w8 = m->host[p + "weight.pad8"];
        w8.assign((size_t)md.cout * 8 * 9, 0.f);
        for (int o = 0; o < md.cout; ++o)
          for (int c = 0; c < 4; ++c)
            for (int k = 0; k < 9; ++k) w8[((size_t)o * 8 + c) * 9 + k] = w4[((size_t)o * 4 + c) * 9 + k];
        FD_TRY(pack_conv(m, p + "weight.pad8", md.cout, 8, 0, 3, "", 0, 0, &md.w0, st));
        FD_TRY(upload_f32(m, p + "bias", &md.b_f32));
        FD_TRY(upload_f32(m, p + "weight", &md.w_f32));   // [Cout][4][3][3] for the vector-FMA kernel of the bf16 mode
        break;
      }
      case M_COMBINE: {
        const std::string q = md.kind == M_COMBINE ? p + "Conv_0." : p;
        FD_TRY(upload_f32(m, q + "weight", &md.w_f32));
        FD_TRY(upload_f32(m, q + "bias", &md.b_f32));
        break;
      }
      case M_GN:
        FD_TRY(upload_f32(m, p + "weight", &md.gn0_g));
        FD_TRY(upload_f32(m, p + "bias", &md.gn0_b));
        break;
      case M_CONV_HEAD:
        FD_TRY(pack_conv(m, p + "weight", md.cout, md.cin, 0, 3, "", 0, 0, &md.w0, st));
        FD_TRY(upload_f32(m, p + "bias", &md.b_f32));
        break;
      case M_RB: {
        FD_TRY(upload_f32(m, p + "GroupNorm_0.weight", &md.gn0_g)); FD_TRY(upload_f32(m, p + "GroupNorm_0.bias", &md.gn0_b));
        FD_TRY(upload_f32(m, p + "GroupNorm_1.weight", &md.gn1_g)); FD_TRY(upload_f32(m, p + "GroupNorm_1.bias", &md.gn1_b));
        // up/down blocks resample the (single) input first, so Conv_0 / Conv_2 see one tensor of cin channels
        // algorithm per convolution: Winograd F(2,3) where the configuration asks for it and the shape is supported
        const bool want_wino = (m->cfg.act_dtype & FD_WINOGRAD) || ((m->cfg.act_dtype & FD_WINOGRAD_LOWRES) && md.level >= 2);
        md.wino0 = want_wino && fd_conv_packed_bytes(md.cout, md.c0, md.c1, 3, 0, 0, m->dt | FD_WINOGRAD) > 0;
        md.wino1 = want_wino && fd_conv_packed_bytes(md.cout, md.cout, 0, 3, md.has_c2 ? md.c0 : 0, md.has_c2 ? md.c1 : 0, m->dt | FD_WINOGRAD) > 0;
        const bool both = (m->cfg.act_dtype & (FD_WINOGRAD_AUTO | FD_LOW_LATENCY)) != 0;   // both packings; the kernel is chosen per launch by its grid
        if (both && fd_conv_packed_bytes(md.cout, md.c0, md.c1, 3, 0, 0, m->dt | FD_WINOGRAD) > 0)
          FD_TRY(pack_conv(m, p + "Conv_0.weight", md.cout, md.c0, md.c1, 3, "", 0, 0, &md.w0w, st, FD_WINOGRAD));
        if (both && fd_conv_packed_bytes(md.cout, md.cout, 0, 3, md.has_c2 ? md.c0 : 0, md.has_c2 ? md.c1 : 0, m->dt | FD_WINOGRAD) > 0)
          FD_TRY(pack_conv(m, p + "Conv_1.weight", md.cout, md.cout, 0, 3, md.has_c2 ? p + "Conv_2.weight" : std::string(), md.has_c2 ? md.c0 : 0,
                           md.has_c2 ? md.c1 : 0, &md.w1w, st, FD_WINOGRAD));
        const bool both4 = (m->cfg.act_dtype & (FD_WINOGRAD_AUTO | FD_LOW_LATENCY)) != 0;
        // fp32 mode: the 2-D float32 kernel's packing wherever its shape rules hold; the F(4,3) float32 packing (whose rules are a subset:
        // Cout = 256, channels % 16) only where they do not -- Fwd::conv prefers the 2-D kernel, a second copy would be dead memory
        if (both4 && m->dt == FD_F32 && !(m->cfg.act_dtype & (FD_BF16_OPERANDS | FD_BF16X3_OPERANDS))) {
          if (fd_conv_packed_bytes(md.cout, md.c0, md.c1, 3, 0, 0, FD_F32 | FD_WINOGRAD44) > 0)
            FD_TRY(pack_conv(m, p + "Conv_0.weight", md.cout, md.c0, md.c1, 3, "", 0, 0, &md.w0w44, st, FD_WINOGRAD44));
          if (fd_conv_packed_bytes(md.cout, md.cout, 0, 3, md.has_c2 ? md.c0 : 0, md.has_c2 ? md.c1 : 0, FD_F32 | FD_WINOGRAD44) > 0)
            FD_TRY(pack_conv(m, p + "Conv_1.weight", md.cout, md.cout, 0, 3, md.has_c2 ? p + "Conv_2.weight" : std::string(), md.has_c2 ? md.c0 : 0,
                             md.has_c2 ? md.c1 : 0, &md.w1w44, st, FD_WINOGRAD44));
        }
        if (both4 && !md.w0w44 && fd_conv_packed_bytes(md.cout, md.c0, md.c1, 3, 0, 0, m->dt | FD_WINOGRAD4) > 0)
          FD_TRY(pack_conv(m, p + "Conv_0.weight", md.cout, md.c0, md.c1, 3, "", 0, 0, &md.w0w4, st, FD_WINOGRAD4));
        if (both4 && !md.w1w44 && fd_conv_packed_bytes(md.cout, md.cout, 0, 3, md.has_c2 ? md.c0 : 0, md.has_c2 ? md.c1 : 0, m->dt | FD_WINOGRAD4) > 0)
          FD_TRY(pack_conv(m, p + "Conv_1.weight", md.cout, md.cout, 0, 3, md.has_c2 ? p + "Conv_2.weight" : std::string(), md.has_c2 ? md.c0 : 0,
                           md.has_c2 ? md.c1 : 0, &md.w1w4, st, FD_WINOGRAD4));
        FD_TRY(pack_conv(m, p + "Conv_0.weight", md.cout, md.c0, md.c1, 3, "", 0, 0, &md.w0, st, md.wino0 ? FD_WINOGRAD : 0));
        if (md.has_c2) {  // fold the 1x1 shortcut into Conv_1's K loop; biases add
          FD_TRY(pack_conv(m, p + "Conv_1.weight", md.cout, md.cout, 0, 3, p + "Conv_2.weight", md.c0, md.c1, &md.w1, st, md.wino1 ? FD_WINOGRAD : 0));
          std::vector<float>& b1 = m->host[p + "Conv_1.bias"];
          const std::vector<float>& b2 = m->host[p + "Conv_2.bias"];
          for (size_t i = 0; i < b1.size(); ++i) b1[i] += b2[i];
        } else {
          FD_TRY(pack_conv(m, p + "Conv_1.weight", md.cout, md.cout, 0, 3, "", 0, 0, &md.w1, st, md.wino1 ? FD_WINOGRAD : 0));
        }
        FD_TRY(upload_f32(m, p + "Conv_1.bias", &md.b1));
        fd_temb_job j{};
        FD_TRY(upload_f32(m, p + "Dense_0.weight", &tmp)); j.dense_w = tmp;
        FD_TRY(upload_f32(m, p + "Dense_0.bias", &tmp)); j.dense_b = tmp;
        FD_TRY(upload_f32(m, p + "Conv_0.bias", &tmp)); j.conv_b = tmp;
        md.bias0_eff = bias_pool + bias_off; bias_off += (size_t)fd_model::MAX_NT * md.cout;
        j.out = md.bias0_eff; j.Cout = md.cout;
        jobs.push_back(j);
        break;
      }
      default: break;
    }
  }
  m->njobs = (int)jobs.size();
  FD_HIP(hipMalloc(&m->jobs_dev, sizeof(fd_temb_job) * jobs.size()));
  m->dev_allocs.push_back(m->jobs_dev);
  FD_HIP(hipMemcpy(m->jobs_dev, jobs.data(), sizeof(fd_temb_job) * jobs.size(), hipMemcpyHostToDevice));
  FD_HIP(hipMalloc(&m->temb, sizeof(float) * fd_model::MAX_NT * m->temb_dim));
  m->dev_allocs.push_back(m->temb);
  if (m->sigma_host.empty()) { m->sigma_host.assign(1, 0.0); m->sigma_n = 1; }
  FD_HIP(hipMalloc(&m->sigma_dev, sizeof(double) * m->n_freq));
  m->dev_allocs.push_back(m->sigma_dev);
  FD_HIP(hipMemcpy(m->sigma_dev, m->sigma_host.data(), sizeof(double) * m->sigma_n, hipMemcpyHostToDevice));
  FD_TRY(fd_stft_plan_create(m->cfg.n_fft, m->cfg.hop, &m->stft));
  if (!(m->cfg.act_dtype & FD_NO_SIDE_STREAM)) {   // (a config bit, not process state: FD_NO_SIDE_STREAM keeps everything on the caller's stream)
    FD_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
    FD_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    FD_HIP(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
  }
  FD_HIP(hipStreamSynchronize(st));
  m->host.clear();
  m->finalized = true;
  return FD_OK;
}

extern "C" size_t fd_model_workspace_bytes(const fd_model* m, int B, int T_pad) {
  if (!m || check_shape(m, B, T_pad) != FD_OK) return 0;
  return ode_ws_bytes(m, B, T_pad);
}

extern "C" int fd_ncsnpp_forward(fd_model* m, const float* x, const float* y, const float* t, int nt, float* v, int B, int T_pad, void* ws,
                                 size_t ws_bytes, void* stream) {
  FD_MODEL_ENTER(m, "fd_ncsnpp_forward");
  FD_TRY(check_ready(m));
  FD_REQUIRE(x && y && t && v && ws, "fd_ncsnpp_forward: null pointer");
  FD_TRY(check_shape(m, B, T_pad));
  FD_REQUIRE(nt == 1 || nt == B, "fd_ncsnpp_forward: t must have 1 or B entries (got %d)", nt);
  FD_REQUIRE(nt <= fd_model::MAX_NT, "fd_ncsnpp_forward: per-sample t supports at most %d clips", fd_model::MAX_NT);
  const size_t need = forward_ws_bytes(m, B, T_pad);
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "fd_ncsnpp_forward: workspace %zu < required %zu bytes", ws_bytes, need);
  OutSpec os; os.dst = v; os.coef = 1.f;
  return forward_call(m, x, y, t, 0.f, nt, os, B, T_pad, ws, ws_bytes, fd_stream(stream));
}

extern "C" int fd_ode_solve(fd_model* m, const float* Y, const float* noise, float sigma_fac, int N, int solver, float* X_out, float* traj, int B,
                            int T_pad, void* ws, size_t ws_bytes, int use_graph, void* stream) {
  FD_MODEL_ENTER(m, "fd_ode_solve");
  FD_TRY(check_ready(m));
  FD_REQUIRE(Y && noise && X_out && ws, "fd_ode_solve: null pointer");
  FD_REQUIRE(N >= 1, "fd_ode_solve: N must be >= 1");
  FD_REQUIRE(solver_nfe(solver, N) > 0, "fd_ode_solve: unknown solver id %d", solver);
  FD_TRY(check_shape(m, B, T_pad));
  const size_t need = ode_ws_bytes(m, B, T_pad);
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "fd_ode_solve: workspace %zu < required %zu bytes", ws_bytes, need);
  hipStream_t st = fd_stream(stream);
  GraphKey key; key.Y = Y; key.noise = noise; key.X = X_out; key.traj = traj; key.ws = ws; key.B = B; key.T = T_pad; key.N = N; key.solver = solver;
  key.kind = 1; key.sigma_fac = sigma_fac;
  return run_maybe_graph(m, key, use_graph != 0, st, [&]() { return ode_enqueue(m, Y, noise, sigma_fac, N, solver, X_out, traj, B, T_pad, ws, ws_bytes, st); });
}

// ---- adaptive Dormand-Prince 5(4) (torchdyn 'dopri5' semantics restated, see oracle odeint_dopri5; host-driven: one
// synchronisation per attempted step for the error ratio) --------------------------------------------------------------
namespace {
constexpr int DP_NORM_BLOCKS = 512;
const double DP_C[7] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
const double DP_A[7][6] = {{0}, {1.0 / 5}, {3.0 / 40, 9.0 / 40}, {44.0 / 45, -56.0 / 15, 32.0 / 9},
                           {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
                           {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656},
                           {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
// error weights b5 - b4 (the 5th-order weights b5 are the last row of DP_A; its 7th entry is 0)
const double DP_E[7] = {35.0 / 384 - 5179.0 / 57600, 0.0, 500.0 / 1113 - 7571.0 / 16695, 125.0 / 192 - 393.0 / 640,
                        -2187.0 / 6784 + 92097.0 / 339200, 11.0 / 84 - 187.0 / 2100, -1.0 / 40};
// Tsitouras 5(4) (torchdyn's `Tsitouras45` / `construct_tsit5`; coefficients as published, verified against the order conditions in
// tests/test_oracle_golden.py): the same 7-stage FSAL shape, E = the published error weights
const double TS_C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
const double TS_A[7][6] = {{0}, {0.161}, {-0.008480655492356989, 0.335480655492357}, {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
                           {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
                           {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383},
                           {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};
const double TS_E[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629, 0.5823571654525552,
                        -0.45808210592918697, 0.015151515151515152};
size_t adaptive_ws_bytes(const fd_model* m, int B, int T) {
  const size_t state = fd_align(sizeof(float) * 2 * (size_t)B * m->n_freq * T);
  return 10 * state + fd_align(sizeof(double) * DP_NORM_BLOCKS) + forward_ws_bytes(m, B, T);
}
}  // namespace

extern "C" size_t fd_ode_adaptive_workspace_bytes(const fd_model* m, int B, int T_pad) {
  if (!m || check_shape(m, B, T_pad) != FD_OK) return 0;
  return adaptive_ws_bytes(m, B, T_pad);
}

extern "C" int fd_ode_solve_adaptive_method(fd_model* m, const float* Y, const float* noise, float sigma_fac, int N, int method, float atol, float rtol,
                                            float* X_out, float* traj, int* nfe_out, int B, int T_pad, void* ws, size_t ws_bytes, void* stream);
extern "C" int fd_ode_solve_adaptive(fd_model* m, const float* Y, const float* noise, float sigma_fac, int N, float atol, float rtol, float* X_out,
                                     float* traj, int* nfe_out, int B, int T_pad, void* ws, size_t ws_bytes, void* stream) {
  return fd_ode_solve_adaptive_method(m, Y, noise, sigma_fac, N, FD_ADAPTIVE_DOPRI5, atol, rtol, X_out, traj, nfe_out, B, T_pad, ws, ws_bytes, stream);
}

extern "C" int fd_ode_solve_adaptive_method(fd_model* m, const float* Y, const float* noise, float sigma_fac, int N, int method, float atol, float rtol,
                                            float* X_out, float* traj, int* nfe_out, int B, int T_pad, void* ws, size_t ws_bytes, void* stream) {
  FD_MODEL_ENTER(m, "fd_ode_solve_adaptive");
  FD_REQUIRE(method == FD_ADAPTIVE_DOPRI5 || method == FD_ADAPTIVE_TSIT5, "fd_ode_solve_adaptive: unknown method id %d", method);
  const double* const TB_C = method == FD_ADAPTIVE_TSIT5 ? TS_C : DP_C;
  const double (*const TB_A)[6] = method == FD_ADAPTIVE_TSIT5 ? TS_A : DP_A;
  const double* const TB_E = method == FD_ADAPTIVE_TSIT5 ? TS_E : DP_E;
  FD_TRY(check_ready(m));
  FD_REQUIRE(Y && noise && X_out && ws, "fd_ode_solve_adaptive: null pointer");
  FD_REQUIRE(N >= 1 && atol > 0.f && rtol >= 0.f, "fd_ode_solve_adaptive: need N >= 1, atol > 0, rtol >= 0");
  FD_TRY(check_shape(m, B, T_pad));
  const size_t need = adaptive_ws_bytes(m, B, T_pad);
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "fd_ode_solve_adaptive: workspace %zu < required %zu bytes", ws_bytes, need);
  hipStream_t st = fd_stream(stream);
  const size_t nstate = (size_t)B * m->n_freq * T_pad;
  const size_t state = fd_align(sizeof(float) * 2 * nstate);
  char* base = (char*)ws;
  float* x = (float*)base; float* x_new = (float*)(base + state); float* err = (float*)(base + 2 * state);
  float* k[7];
  for (int i = 0; i < 7; ++i) k[i] = (float*)(base + (3 + i) * state);
  double* partial = (double*)(base + 10 * state);
  void* fws = base + 10 * state + fd_align(sizeof(double) * DP_NORM_BLOCKS);
  const size_t fws_bytes = ws_bytes - (10 * state + fd_align(sizeof(double) * DP_NORM_BLOCKS));
  std::vector<double> host_partial(DP_NORM_BLOCKS);
  int nfe = 0;
  auto eval = [&](const float* xin, float t, float* kout) {
    OutSpec os; os.dst = kout; os.coef = 1.f;
    ++nfe;
    return forward_call(m, xin, Y, nullptr, t, 1, os, B, T_pad, fws, fws_bytes, st);
  };
  // Hairer norm of (p - q) / (atol + rtol max(|r|, |s|)): sqrt(mean |.|^2) over the complex elements
  auto norm = [&](const float* p_, const float* q_, const float* r_, const float* s_, double* out) {
    FD_TRY(fd_ode_scaled_sq(p_, q_, r_, s_, atol, rtol, partial, DP_NORM_BLOCKS, (long long)nstate, st));
    FD_HIP(hipMemcpyAsync(host_partial.data(), partial, sizeof(double) * DP_NORM_BLOCKS, hipMemcpyDeviceToHost, st));
    FD_HIP(hipStreamSynchronize(st));
    double acc = 0.0;
    for (double v : host_partial) acc += v;
    *out = sqrt(acc / (double)nstate);
    return FD_OK;
  };
  FD_TRY(fd_init_state(Y, noise, m->sigma_dev, m->sigma_n, sigma_fac, x, B, m->n_freq, T_pad, st));
  if (traj) FD_HIP(hipMemcpyAsync(traj, x, sizeof(float) * 2 * nstate, hipMemcpyDeviceToDevice, st));
  const std::vector<float> ts = t_span_linspace(N);
  float t = ts[0];
  const float T = ts[N];
  FD_TRY(eval(x, t, k[0]));
  // initial step (Hairer): h0 = 0.01 d0 / d1, one explicit Euler probe, h1 = (0.01 / max(d1, d2))^(1/6)
  double d0, d1, d2;
  FD_TRY(norm(x, nullptr, x, x, &d0));
  FD_TRY(norm(k[0], nullptr, x, x, &d1));
  const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
  {
    const float* kk[1] = {k[0]}; const float cc[1] = {1.f};
    FD_TRY(fd_ode_lincomb(x, 1.f, (float)h0, kk, cc, 1, x_new, (long long)nstate, st));
    FD_TRY(eval(x_new, t + (float)h0, k[1]));
    FD_TRY(norm(k[1], k[0], x, x, &d2));
    d2 /= h0;
  }
  const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / 6.0);
  float dt = (float)fmin(100.0 * h0, h1);
  int ckpt = 0, steps = 0;
  while (t < T) {
    if (++steps > 100000) return fd_set_error(FD_ERUNTIME, "fd_ode_solve_adaptive: step limit reached (dt = %g at t = %g)", (double)dt, (double)t);
    if (t + dt > T) dt = T - t;
    bool flag = false;
    float dt_old = dt;
    if (ckpt < N && t + dt > ts[ckpt + 1]) { dt_old = dt; flag = true; dt = ts[ckpt + 1] - t; }
    for (int s = 1; s < 7; ++s) {                                    // stages 2..7; stage 7 is evaluated at the 5th-order solution (FSAL)
      const float* kk[6]; float cc[6];
      for (int j = 0; j < s; ++j) { kk[j] = k[j]; cc[j] = (float)TB_A[s][j]; }
      float* xs = s == 6 ? x_new : err;                              // `err` doubles as the stage input buffer
      FD_TRY(fd_ode_lincomb(x, 1.f, dt, kk, cc, s, xs, (long long)nstate, st));
      FD_TRY(eval(xs, t + (float)TB_C[s] * dt, k[s]));
    }
    {
      const float* kk[7]; float cc[7];
      for (int j = 0; j < 7; ++j) { kk[j] = k[j]; cc[j] = (float)TB_E[j]; }
      FD_TRY(fd_ode_lincomb(x, 0.f, dt, kk, cc, 7, err, (long long)nstate, st));
    }
    double ratio;
    FD_TRY(norm(err, nullptr, x, x_new, &ratio));
    if (ratio <= 1.0) {
      t = t + dt;
      std::swap(x, x_new);
      std::swap(k[0], k[6]);
      if (ckpt < N && fabs((double)t - (double)ts[ckpt + 1]) <= 1e-7) {
        t = ts[ckpt + 1];
        ++ckpt;
        if (traj) FD_HIP(hipMemcpyAsync(traj + 2 * nstate * ckpt, x, sizeof(float) * 2 * nstate, hipMemcpyDeviceToDevice, st));
      }
    }
    if (flag) dt = dt_old - dt;
    if (ratio == 0.0) dt = dt * 10.f;
    else {
      const double minf = ratio < 1.0 ? 1.0 : 0.2;
      dt = (float)((double)dt * fmin(10.0, fmax(0.9 / pow(ratio, 1.0 / 5.0), minf)));
    }
  }
  FD_HIP(hipMemcpyAsync(X_out, x, sizeof(float) * 2 * nstate, hipMemcpyDeviceToDevice, st));
  FD_HIP(hipStreamSynchronize(st));
  if (nfe_out) *nfe_out = nfe;
  return FD_OK;
}

extern "C" size_t fd_enhance_workspace_bytes(const fd_model* m, int B, int L) {
  if (!m || B <= 0 || L <= 0) return 0;
  const int T = 1 + L / m->cfg.hop, Tp = fd_padded_frames(T);
  if (check_shape(m, B, Tp) != FD_OK) return 0;
  const size_t state = fd_align(sizeof(float) * 2 * (size_t)B * m->n_freq * Tp);
  const size_t a = fd_stft_ws_bytes(B, L, m->cfg.n_fft, m->cfg.hop), b = ode_ws_bytes(m, B, Tp);
  return 2 * state + fd_align(sizeof(float) * B) + (a > b ? a : b) + 256;
}

extern "C" size_t fd_enhance_normfac_offset(const fd_model* m, int B, int L) {
  if (!m || B <= 0 || L <= 0) return 0;
  const int Tp = fd_padded_frames(1 + L / m->cfg.hop);
  return 2 * fd_align(sizeof(float) * 2 * (size_t)B * m->n_freq * Tp);
}

namespace {
int enhance_impl(fd_model* m, const char* who, const float* y, const int* lens, const float* noise, float sigma_fac, int N, int solver, float* x_hat, int B,
                 int L, void* ws, size_t ws_bytes, int use_graph, void* stream) {
  FD_TRY(check_ready(m));
  FD_REQUIRE(y && noise && x_hat && ws, "%s: null pointer", who);
  FD_REQUIRE(N >= 1 && solver_nfe(solver, N) > 0, "%s: bad N / solver", who);
  FD_REQUIRE(B > 0 && L > m->cfg.n_fft / 2, "%s: clips must be longer than %d samples", who, m->cfg.n_fft / 2);
  const int T = 1 + L / m->cfg.hop, Tp = fd_padded_frames(T);
  FD_TRY(check_shape(m, B, Tp));
  const size_t need = fd_enhance_workspace_bytes(m, B, L);
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "%s: workspace %zu < required %zu bytes", who, ws_bytes, need);
  hipStream_t st = fd_stream(stream);
  const size_t state = fd_align(sizeof(float) * 2 * (size_t)B * m->n_freq * Tp);
  float* Y = (float*)ws;
  float* X = (float*)((char*)ws + state);
  float* normfac = (float*)((char*)ws + 2 * state);
  char* rest = (char*)ws + 2 * state + fd_align(sizeof(float) * B);
  const size_t rest_bytes = ws_bytes - (2 * state + fd_align(sizeof(float) * B));
  GraphKey key; key.y = y; key.noise = noise; key.xhat = x_hat; key.ws = ws; key.lens = lens; key.B = B; key.L = L; key.N = N; key.solver = solver; key.kind = 2;
  key.sigma_fac = sigma_fac; key.normalize = m->normalize;
  return run_maybe_graph(m, key, use_graph != 0, st, [&]() {
    FD_TRY(fd_stft_forward(m->stft, y, lens, B, L, m->cfg.alpha, m->cfg.beta, m->normalize, normfac, Y, Tp, rest, rest_bytes, st));
    FD_TRY(ode_enqueue(m, Y, noise, sigma_fac, N, solver, X, nullptr, B, Tp, rest, rest_bytes, st));
    FD_TRY(fd_stft_inverse(m->stft, X, lens, B, T, Tp, m->cfg.alpha, m->cfg.beta, normfac, x_hat, L, rest, rest_bytes, st));
    return FD_OK;
  });
}
}  // namespace

extern "C" int fd_enhance(fd_model* m, const float* y, const float* noise, float sigma_fac, int N, int solver, float* x_hat, int B, int L, void* ws,
                          size_t ws_bytes, int use_graph, void* stream) {
  FD_MODEL_ENTER(m, "fd_enhance");
  return enhance_impl(m, "fd_enhance", y, nullptr, noise, sigma_fac, N, solver, x_hat, B, L, ws, ws_bytes, use_graph, stream);
}

// FlowModel.enhance on a RAGGED batch: the reference's driver enhances a directory file by file (enhance.py:96-137), every file its
// own length; here the files whose spectrograms pad to the same T_pad (util/other.py:25-52) run as ONE batch.  y / x_hat are [B][L]
// rows (L = the longest clip; workspace as for fd_enhance(B, L)), lengths (device int32 [B]) the clips' own sample counts -- the
// CALLER guarantees fd_padded_frames(fd_num_frames(lengths[b])) == fd_padded_frames(fd_num_frames(L)) for every b (the kernels
// clamp a length into (n_fft/2, L]).  Clip b's result is bit-identical to fd_enhance on that clip alone with the same noise
// (noise is [B][1][F][T_pad] as usual); samples [lengths[b], L) of a row of x_hat are zero.  The hipGraph of a (B, L) bucket is
// keyed on the POINTER `lengths`: its contents may change between replays.
extern "C" int fd_enhance_ragged(fd_model* m, const float* y, const int* lengths, const float* noise, float sigma_fac, int N, int solver,
                                 float* x_hat, int B, int L, void* ws, size_t ws_bytes, int use_graph, void* stream) {
  FD_MODEL_ENTER(m, "fd_enhance_ragged");
  FD_REQUIRE(lengths, "fd_enhance_ragged: null lengths");
  return enhance_impl(m, "fd_enhance_ragged", y, lengths, noise, sigma_fac, N, solver, x_hat, B, L, ws, ws_bytes, use_graph, stream);
}

// Shared front end / back end of the three enhancement models:  STFT -> body(Y, X) -> iSTFT
template <typename Body>
static int enhance_common(fd_model* m, const char* who, const float* y, float* x_hat, int B, int L, void* ws, size_t ws_bytes, GraphKey key,
                          int use_graph, void* stream, Body&& body) {
  FD_TRY(check_ready(m));
  FD_REQUIRE(y && x_hat && ws, "%s: null pointer", who);
  FD_REQUIRE(B > 0 && L > m->cfg.n_fft / 2, "%s: clips must be longer than %d samples", who, m->cfg.n_fft / 2);
  const int T = 1 + L / m->cfg.hop, Tp = fd_padded_frames(T);
  FD_TRY(check_shape(m, B, Tp));
  const size_t need = fd_enhance_workspace_bytes(m, B, L);
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "%s: workspace %zu < required %zu bytes", who, ws_bytes, need);
  hipStream_t st = fd_stream(stream);
  const size_t state = fd_align(sizeof(float) * 2 * (size_t)B * m->n_freq * Tp);
  float* Y = (float*)ws;
  float* X = (float*)((char*)ws + state);
  float* normfac = (float*)((char*)ws + 2 * state);
  char* rest = (char*)ws + 2 * state + fd_align(sizeof(float) * B);
  const size_t rest_bytes = ws_bytes - (2 * state + fd_align(sizeof(float) * B));
  key.y = y; key.xhat = x_hat; key.ws = ws; key.B = B; key.L = L; key.normalize = m->normalize;
  return run_maybe_graph(m, key, use_graph != 0, st, [&]() {
    FD_TRY(fd_stft_forward(m->stft, y, nullptr, B, L, m->cfg.alpha, m->cfg.beta, m->normalize, normfac, Y, Tp, rest, rest_bytes, st));
    FD_TRY(body(Y, X, Tp, (void*)rest, rest_bytes, st));
    FD_TRY(fd_stft_inverse(m->stft, X, nullptr, B, T, Tp, m->cfg.alpha, m->cfg.beta, normfac, x_hat, L, rest, rest_bytes, st));
    return FD_OK;
  });
}

extern "C" int fd_score_num_draws(const fd_score_config* c) {
  if (!c || c->N < 1 || c->corrector_steps < 0) return -1;
  return score_draws(*c);
}

extern "C" int fd_score_enhance(fd_model* m, const float* y, const float* noise, const fd_score_config* c, float* x_hat, int B, int L, void* ws,
                                size_t ws_bytes, int use_graph, void* stream) {
  FD_MODEL_ENTER(m, "fd_score_enhance");
  FD_REQUIRE(c && noise, "fd_score_enhance: null pointer");
  FD_REQUIRE(c->N >= 1, "fd_score_enhance: N must be >= 1");
  FD_REQUIRE(c->predictor >= FD_PREDICTOR_REVERSE_DIFFUSION && c->predictor <= FD_PREDICTOR_NONE, "fd_score_enhance: unknown predictor id %d", c->predictor);
  FD_REQUIRE(c->corrector == FD_CORRECTOR_ALD || c->corrector == FD_CORRECTOR_NONE, "fd_score_enhance: unknown corrector id %d", c->corrector);
  FD_REQUIRE(c->corrector_steps >= 0 && c->corrector_steps <= 64, "fd_score_enhance: corrector_steps out of range");
  FD_REQUIRE(c->sigma_min > 0.f && c->sigma_max > c->sigma_min && c->theta > 0.f, "fd_score_enhance: bad OUVE parameters");
  FD_REQUIRE(c->t_eps > 0.f && c->t_eps < 1.f, "fd_score_enhance: t_eps must be in (0, 1)");
  GraphKey key; key.noise = noise; key.kind = 3; key.score = *c;
  const fd_score_config cfg = *c;
  return enhance_common(m, "fd_score_enhance", y, x_hat, B, L, ws, ws_bytes, key, use_graph, stream,
                        [&](float* Y, float* X, int Tp, void* rest, size_t rest_bytes, hipStream_t st) {
                          return score_enqueue(m, Y, noise, cfg, X, B, Tp, rest, rest_bytes, st);
                        });
}

// One evaluation of the score network combined into what the black-box ODE sampler of the reference needs
// (sampling/__init__.py:75-146): the probability-flow drift rsde.sde(x, t, y)[0] (sdes.py:93-109) or the noise-free
// reverse-diffusion predictor step used for the final denoising (predictors.py:61-71 with t = eps).
extern "C" int fd_score_eval(fd_model* m, const float* x, const float* Y, float t, const fd_score_config* c, int mode, float* out, int B,
                             int T_pad, void* ws, size_t ws_bytes, void* stream) {
  FD_MODEL_ENTER(m, "fd_score_eval");
  FD_TRY(check_ready(m));
  FD_REQUIRE(x && Y && c && out && ws, "fd_score_eval: null pointer");
  FD_REQUIRE(mode >= FD_SCORE_DRIFT_PF && mode <= FD_SCORE_DENOISE, "fd_score_eval: unknown mode %d", mode);
  FD_REQUIRE(c->sigma_min > 0.f && c->sigma_max > c->sigma_min && c->theta > 0.f, "fd_score_eval: bad OUVE parameters");
  FD_REQUIRE(mode != FD_SCORE_DENOISE || c->N >= 1, "fd_score_eval: the denoising step needs N >= 1");
  FD_TRY(check_shape(m, B, T_pad));
  const size_t need = forward_ws_bytes(m, B, T_pad);
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "fd_score_eval: workspace %zu < required %zu bytes", ws_bytes, need);
  const float std_t = ouve_std(*c, t), g = ouve_diffusion(*c, t);
  OutSpec os; os.score = true; os.base = x; os.yv = Y; os.dst = out;
  if (mode == FD_SCORE_DENOISE) {   // x_mean = x - (theta (y - x) dt - G^2 score), G = g sqrt(dt), score = -net / std
    const float dt = 1.f / (float)c->N, G = g * sqrtf(dt);
    os.cb = 1.f + c->theta * dt; os.cy = -c->theta * dt; os.coef = -(G * G) / std_t;
  } else {                          // drift = theta (y - x) - g^2 score * (0.5 | 1)
    os.cb = -c->theta; os.cy = c->theta; os.coef = (mode == FD_SCORE_DRIFT_PF ? 0.5f : 1.f) * g * g / std_t;
  }
  return forward_call(m, x, Y, nullptr, t, 1, os, B, T_pad, ws, ws_bytes, fd_stream(stream));
}

extern "C" int fd_regression_enhance(fd_model* m, const float* y, float* x_hat, int B, int L, void* ws, size_t ws_bytes, int use_graph,
                                     void* stream) {
  FD_MODEL_ENTER(m, "fd_regression_enhance");
  GraphKey key; key.kind = 4;
  return enhance_common(m, "fd_regression_enhance", y, x_hat, B, L, ws, ws_bytes, key, use_graph, stream,
                        [&](float* Y, float* X, int Tp, void* rest, size_t rest_bytes, hipStream_t st) {
                          OutSpec os; os.dst = X; os.coef = 1.f;                                 // X_hat = backbone(Y, Y, t = 0), model.py:566-578
                          return forward_call(m, Y, Y, nullptr, 0.f, 1, os, B, Tp, rest, rest_bytes, st);
                        });
}

// ---- one ResnetBlockBigGANpp as a stand-alone call (layerspp.py:252-284; the fusion unit of SURVEY 8(a13)) -----------
namespace {
int resblock_run(const fd_resblock_desc& d, const void* x0, const void* x1, void* out, int B, int H, int W, int dtype, bool dry, void* ws,
                 hipStream_t st, size_t* peak) {
  Mod md; md.kind = M_RB; md.cin = d.cin0 + d.cin1; md.c0 = d.cin0; md.c1 = d.cin1; md.cout = d.cout;
  md.up = d.up != 0; md.down = d.down != 0; md.has_c2 = d.has_conv2 != 0;
  md.w0 = const_cast<void*>(d.w0); md.w1 = const_cast<void*>(d.w1);
  md.gn0_g = const_cast<float*>(d.gn0_gamma); md.gn0_b = const_cast<float*>(d.gn0_beta);
  md.gn1_g = const_cast<float*>(d.gn1_gamma); md.gn1_b = const_cast<float*>(d.gn1_beta);
  md.bias0_eff = const_cast<float*>(d.bias0); md.b1 = const_cast<float*>(d.bias1);
  Fwd f{nullptr, dry, (char*)ws, Arena(), st, B, 0, 0, dtype, (int)fd_dtype_size(dtype)};
  // tensors that live outside the workspace are addressed relative to it (never released to the arena)
  auto external = [&](const void* p, int C, int h, int w) { Tens t; t.off = (size_t)((const char*)p - (const char*)ws); t.C = C; t.H = h; t.W = w; return t; };
  Tens a = external(dry ? ws : x0, d.cin0, H, W), b, o = external(dry ? ws : out, d.cout, 0, 0);
  if (d.cin1) b = external(dry ? ws : x1, d.cin1, H, W);
  const int rc = f.resblock(md, a, d.cin1 ? &b : nullptr, d.bias0_rows, o, true);
  if (peak) *peak = f.arena.peak() + 256;
  return rc;
}
int check_resblock(const fd_resblock_desc* d, int B, int H, int W, int dtype) {
  FD_REQUIRE(d, "fd_resblock: null descriptor");
  FD_REQUIRE(d->cin0 > 0 && d->cin0 % 8 == 0 && d->cin1 >= 0 && d->cin1 % 8 == 0 && d->cout > 0 && d->cout % 8 == 0, "fd_resblock: channel counts must be multiples of 8");
  FD_REQUIRE(!(d->up && d->down) && (!(d->up || d->down) || d->cin1 == 0), "fd_resblock: up / down blocks take a single input tensor");
  FD_REQUIRE(d->has_conv2 || d->cin0 + d->cin1 == d->cout, "fd_resblock: identity shortcut needs Cin == Cout");
  FD_REQUIRE(B > 0 && H > 0 && W > 0 && (!d->down || (H % 2 == 0 && W % 2 == 0)), "fd_resblock: bad shape");
  FD_REQUIRE(dtype == FD_BF16 || dtype == FD_F32, "fd_resblock: bad dtype");
  return FD_OK;
}
}  // namespace

extern "C" size_t fd_resblock_workspace_bytes(const fd_resblock_desc* d, int B, int H, int W, int dtype) {
  if (check_resblock(d, B, H, W, dtype) != FD_OK) return 0;
  size_t peak = 0;
  if (resblock_run(*d, nullptr, nullptr, nullptr, B, H, W, dtype, true, nullptr, nullptr, &peak) != FD_OK) return 0;
  return peak;
}

extern "C" int fd_resblock(const fd_resblock_desc* d, const void* x0, const void* x1, void* out, int B, int H, int W, int dtype, void* ws,
                           size_t ws_bytes, void* stream) {
  FD_TRY(check_resblock(d, B, H, W, dtype));
  FD_REQUIRE(x0 && out && ws && (d->cin1 == 0) == (x1 == nullptr), "fd_resblock: null pointer / x1 mismatch");
  FD_REQUIRE(d->w0 && d->w1 && d->gn0_gamma && d->gn0_beta && d->gn1_gamma && d->gn1_beta && d->bias0 && d->bias1, "fd_resblock: incomplete descriptor");
  FD_REQUIRE(d->bias0_rows == 1 || d->bias0_rows == B, "fd_resblock: bias0_rows must be 1 or B");
  const size_t need = fd_resblock_workspace_bytes(d, B, H, W, dtype);
  if (ws_bytes < need) return fd_set_error(FD_ENOMEM, "fd_resblock: workspace %zu < required %zu bytes", ws_bytes, need);
  return resblock_run(*d, x0, x1, out, B, H, W, dtype, false, ws, fd_stream(stream), nullptr);
}

extern "C" int fd_profile_enable(fd_model* m, int enable) {
  FD_REQUIRE(m, "fd_profile_enable: null model");
  m->profiling = enable != 0;
  m->ev_used = 0;
  m->prof_flops = 0.0;
  m->prof_flops_exec = 0.0;
  m->prof_bytes = 0.0;
  m->ev_fir_used = 0;
  m->prof_fir_bytes = 0.0;
  if (m->stft) FD_TRY(fd_stft_plan_profile(m->stft, enable));
  return FD_OK;
}

extern "C" int fd_profile_read_stft(fd_model* m, double* ms6, int* calls2) {
  FD_REQUIRE(m && m->stft, "fd_profile_read_stft: null model / model not finalised");
  return fd_stft_plan_profile_read(m->stft, ms6, calls2);
}

extern "C" int fd_profile_read_fir(fd_model* m, double* ms_total, long long* launches, double* bytes_total) {
  FD_REQUIRE(m, "fd_profile_read_fir: null model");
  double ms = 0.0;
  for (size_t i = 0; i < m->ev_fir_used; ++i) {
    FD_HIP(hipEventSynchronize(m->ev_fir[i].second));
    float e = 0.f;
    FD_HIP(hipEventElapsedTime(&e, m->ev_fir[i].first, m->ev_fir[i].second));
    ms += e;
  }
  if (ms_total) *ms_total = ms;
  if (launches) *launches = (long long)m->ev_fir_used;
  if (bytes_total) *bytes_total = m->prof_fir_bytes;
  m->ev_fir_used = 0;
  m->prof_fir_bytes = 0.0;
  return FD_OK;
}

extern "C" int fd_profile_read(fd_model* m, double* conv_ms_total, long long* conv_launches, double* conv_flops_total, double* conv_bytes_total) {
  FD_REQUIRE(m, "fd_profile_read: null model");
  double ms = 0.0;
  for (size_t i = 0; i < m->ev_used; ++i) {
    FD_HIP(hipEventSynchronize(m->ev[i].second));
    float e = 0.f;
    FD_HIP(hipEventElapsedTime(&e, m->ev[i].first, m->ev[i].second));
    ms += e;
  }
  if (conv_ms_total) *conv_ms_total = ms;
  if (conv_launches) *conv_launches = (long long)m->ev_used;
  if (conv_flops_total) *conv_flops_total = m->prof_flops;
  if (conv_bytes_total) *conv_bytes_total = m->prof_bytes;
  m->prof_bytes = 0.0;
  m->ev_used = 0;
  m->prof_flops = 0.0;
  return FD_OK;
}

extern "C" int fd_profile_read_executed(fd_model* m, double* conv_flops_executed) {
  FD_REQUIRE(m && conv_flops_executed, "fd_profile_read_executed: null argument");
  *conv_flops_executed = m->prof_flops_exec;
  m->prof_flops_exec = 0.0;
  return FD_OK;
}
