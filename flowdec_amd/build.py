"""Builds libflowdec_hip.so (gfx950) in-tree with hipcc: one object per .hip file, compiled in parallel."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libflowdec_hip.so")
SOURCES = ["api.hip", "calib.hip", "conv_mfma.hip", "conv_wino.hip", "conv_wino4.hip", "conv_wino4f.hip", "conv_wino44f.hip", "conv_head.hip", "conv_headf.hip", "elementwise.hip", "stft.hip", "model.hip", "ndac.hip", "ndac_mfma.hip"]
# -fno-slp-vectorize: hipcc (ROCm 7.2) otherwise packs adjacent f32 FMAs into v_pk_fma_f32; beside MFMAs that is slower
# (guide: MI355X_MICROARCH "price of one filler beside MFMAs") and one such packing of the fused GroupNorm affine
# produced wrong lanes (op_sel_hi broadcast) in the f32 conv path.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-I" + INC, "-I" + CSRC, "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INC, "flowdec_hip.h"), os.path.abspath(__file__)]   # (this file: SOURCES / FLAGS)
    return any(os.path.getmtime(d) > t for d in deps)


def check_wino4_isa(hipcc=None, extra=()):
    """conv_wino4.hip (and the float32 kernels conv_wino4f.hip, conv_wino44f.hip) writes M0 from inline asm without saving it (the LDS-DMA destination; hipcc refuses M0 on a clobber list) and counts
    its own s_waitcnt vmcnt by hand.  Both rest on properties of the GENERATED code, so the build checks them and fails otherwise: no M0
    use outside the kernel's own `s_mov_b32 m0` statements, no scratch (a spill inside the K loop would break the counted waits), and the
    expected number of MFMA sites (one loop body per instantiation).  ~6 s, runs beside the object compiles.
    The M0 and scratch properties are hard requirements; the MFMA-site count depends on how a given hipcc unrolls and is a WARNING only
    (a duplicated loop body costs registers, not correctness).  `FLOWDEC_SKIP_ISA_CHECK=1` skips the whole check (another toolchain)."""
    import tempfile
    if os.environ.get("FLOWDEC_SKIP_ISA_CHECK", "") not in ("", "0"):
        print("flowdec_amd.build: FLOWDEC_SKIP_ISA_CHECK set -- conv_wino4.hip ISA properties NOT verified", file=sys.stderr)
        return False
    hipcc = hipcc or _hipcc()
    ok = True
    # file -> (MFMA mnemonic, expected sites): conv_wino4.hip six instantiations x 18 steps x 4; conv_wino4f.hip (float32) six x 18 x 16 in the
    # K loop + 64 per folded-shortcut stage body of the two shortcut instantiations; conv_wino44f.hip (2-D float32) four x (72 per 3x3 chunk
    # body + 32 per shortcut chunk body)
    for src, (mnem, want) in {"conv_wino4.hip": ("v_mfma_f32_32x32x16_f16", 6 * 72), "conv_wino4f.hip": ("v_mfma_f32_32x32x2_f32", 6 * 288 + 2 * 64),
                              "conv_wino44f.hip": ("v_mfma_f32_16x16x4_f32", 4 * (72 + 32))}.items():
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "w4.s")
            r = subprocess.run([hipcc, *FLAGS, *extra, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc -S failed for %s:\n%s" % (src, r.stderr[-4000:]))
            code = [l.split(";")[0] for l in open(out).read().splitlines()]
        foreign = [l for l in code if ("m0" in l.split() or ", m0" in l or " m0," in l) and not l.strip().startswith("s_mov_b32 m0,")]
        if foreign:
            raise RuntimeError("%s: M0 is used outside the kernel's own LDS-DMA statements: %r" % (src, foreign[:5]))
        if any("scratch_" in l for l in code):
            raise RuntimeError("%s: a kernel spills to scratch (breaks the hand-counted s_waitcnt vmcnt)" % src)
        n = sum(mnem in l for l in code)
        if n != want:
            print("flowdec_amd.build: WARNING %s has %d Winograd MFMA sites, expected %d (the K loop was duplicated or unswitched by "
                  "this hipcc: slower, still correct)" % (src, n, want), file=sys.stderr)
            ok = False
    return ok


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    extra = os.environ.get("FLOWDEC_EXTRA_FLAGS", "").split()

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-4000:]))
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES) + 1) as ex:
        isa = ex.submit(check_wino4_isa, hipcc, tuple(extra))
        objs = list(ex.map(cc, SOURCES))
        isa.result()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    if verbose:
        print("built", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
