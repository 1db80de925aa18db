"""Build the gfx950 shared library ``scratchpad_amd/lib/libscratchpad_hip.so`` with hipcc.

Cross-compiles without a GPU.  Objects go to ``build/`` (git-ignored); the ``.so`` stays in-tree so
it travels to the GPU box with the repo snapshot."""
import concurrent.futures
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libscratchpad_hip.so")
OBJ = os.path.join(ROOT, "build", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wno-unused-value"]
# elementwise.hip reproduces torch's per-op rounding (round16(a*b) - round16(c*d)); the default
# -ffp-contract=fast lets the backend fuse that into v_fma_f16 whatever the source pragmas say
# extend_mfma.hip: without -fno-honor-nans every fmaxf on an MFMA result is preceded by a canonicalising
# v_max_f32 x, x (58 instead of 25 vector instructions for a tile's row maximum); nothing in that file
# relies on NaN propagation (masked scores are -inf, never NaN)
# allreduce.hip: its fused all-reduce + add + RMSNorm kernel reproduces elementwise.hip's norm bit for bit
PER_FILE_FLAGS = {"elementwise.hip": ["-ffp-contract=off"], "allreduce.hip": ["-ffp-contract=off"],
                  "extend_mfma.hip": ["-fno-honor-nans"],
                  # extend_w64.hip: every filler of its hand-placed MFMA gaps is a single instruction: no SLP packing
                  # of adjacent f32 adds / multiplies into v_pk_* (MI355X_MICROARCH.md: an anti-lever beside MFMAs).  Its
                  # accumulation registers belong to its assembly text alone: compile_w64 below / tools/w64_asm.py.  The
                  # atomic optimizer would turn the persistent form's ticket draw into atomic + full wait + broadcast.
                  "extend_w64.hip": ["-fno-honor-nans", "-fno-slp-vectorize", "-std=c++20", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _hipcc_version() -> str:
    try:
        out = subprocess.run([HIPCC, "--version"], check=True, capture_output=True, text=True).stdout
    except (OSError, subprocess.CalledProcessError) as e:
        return f"hipcc --version failed: {e}"
    return " | ".join(l.strip() for l in out.splitlines() if l.strip() and not l.startswith("InstalledDir"))


def _w64_asm():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import w64_asm
    finally:
        sys.path.pop(0)
    return w64_asm


def check_w64_descriptors(lib: str = LIB, verbose: bool = False) -> None:
    """Raise unless the library's extend_w64 kernels carry what this source tree expects (tools/w64_asm.py check):
    kernel descriptor == .amdgpu_metadata note == ACCUM_OFFSET + 256 registers of which 256 accumulation, and the host
    flag the launcher checks is 1.  A library linked from a plain `hipcc -c extend_w64.hip`, or compiled by a toolchain
    that lays the kernels' registers out differently, fails the BUILD, not a parity test on a GPU box."""
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ok = _w64_asm().check(lib)
    if verbose or not ok:
        print(buf.getvalue(), end="")
    if not ok:
        raise RuntimeError(f"{lib}: the extend_w64 kernel descriptors are not what extend_w64.hip needs (see above); "
                           "rebuild with `python -m scratchpad_amd.build --force`")


def compile_w64(src: str, obj: str, extra=(), verbose: bool = True) -> str:
    """extend_w64.hip in stages, so that the accumulation registers only its asm text names are allocated where the
    ASSEMBLER can see and validate them (until round 4 the linked .so was byte-patched):
    device assembly -> tools/w64_asm.py rewrite (.amdhsa_next_free_vgpr and the metadata note of the w64 kernels) ->
    clang -x assembler -> lld -> offload bundle -> the host object with that bundle embedded.  These are the commands
    `hipcc -c` runs itself (`hipcc -###`), with the rewrite between its first two."""
    llvm = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin")
    if not os.path.exists(os.path.join(llvm, "lld")):
        llvm = "/opt/rocm/lib/llvm/bin"
    # (intermediates in a directory of their own: variant scripts link build/obj/*.o)
    stage = os.path.join(os.path.dirname(os.path.abspath(obj)), "w64_stage")
    os.makedirs(stage, exist_ok=True)
    base = os.path.join(stage, os.path.basename(obj)[:-2] if obj.endswith(".o") else os.path.basename(obj))
    flags = FLAGS + PER_FILE_FLAGS["extend_w64.hip"] + list(extra) + ["-DSP_W64_REGISTERS_FROM_ASM=1",
                                                                     "-Wno-unused-command-line-argument"]

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    run([HIPCC] + flags + ["--cuda-device-only", "-S", src, "-o", base + ".dev.s"])
    w64 = _w64_asm()
    with open(base + ".dev.w64.s", "w") as f:
        f.write(w64.rewrite(open(base + ".dev.s").read()))
    run([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
         base + ".dev.w64.s", "-o", base + ".dev.o"])
    run([os.path.join(llvm, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o",
         base + ".hsaco", base + ".dev.o"])
    if not w64.check(base + ".hsaco", verbose=verbose):
        raise RuntimeError(f"{base}.hsaco: descriptor / metadata of the w64 kernels are not what the rewrite asked for")
    run([os.path.join(llvm, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null",
         "-input=" + base + ".hsaco", "-output=" + base + ".hipfb"])
    run([HIPCC] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", base + ".hipfb",
                           "-c", src, "-o", obj])
    return obj


def build_native(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(CSRC, "*.inc"))) + [
        os.path.join(ROOT, "include", "scratchpad_hip.h"), os.path.join(ROOT, "tools", "w64_asm.py")]
    # (the library also has to be made of exactly these sources and flags: a source removed - or a flag changed - leaves
    # every remaining file older than the library)
    stamp = LIB + ".sources"
    made_of = "\n".join([os.path.basename(x) for x in srcs] + [" ".join(FLAGS)] +
                        [k + " " + " ".join(v) for k, v in sorted(PER_FILE_FLAGS.items())] +
                        ["compiler: " + _hipcc_version()])
    same = os.path.exists(stamp) and open(stamp).read() == made_of
    if not force and same and _newer(LIB, srcs + hdrs):
        check_w64_descriptors(LIB)       # (cheap: reads the ELF; a library somebody linked by hand fails here)
        return LIB
    force = force or not same
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if not force and _newer(obj, [src] + hdrs):
            return obj
        if os.path.basename(src) == "extend_w64.hip":
            return compile_w64(src, obj, verbose=verbose)
        cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    check_w64_descriptors(LIB, verbose)
    with open(stamp, "w") as f:
        f.write(made_of)
    return LIB


if __name__ == "__main__":
    if "--w64-object" in sys.argv:      # python -m scratchpad_amd.build --w64-object OUT.o [-DFLAG ...]: variant builds
        i = sys.argv.index("--w64-object")
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[i + 1])), exist_ok=True)
        print(compile_w64(os.path.join(CSRC, "extend_w64.hip"), sys.argv[i + 1], extra=sys.argv[i + 2:]))
    else:
        print(build_native(force="--force" in sys.argv))
