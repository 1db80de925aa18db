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
                  # accumulation registers belong to its assembly text alone: see tools/patch_w64_descriptor.py.  The
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


def check_w64_descriptors(lib: str = LIB, verbose: bool = False) -> None:
    """Raise unless the library's extend_w64 kernels carry the patched descriptors this source tree expects
    (tools/patch_w64_descriptor.py --check --expect): an unpatched or differently laid-out library fails the BUILD,
    not a parity test on a GPU box."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "patch_w64_descriptor.py"), lib, "--check", "--expect"],
                       capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout + r.stderr, end="")
    if r.returncode != 0:
        raise RuntimeError(f"{lib}: the extend_w64 kernel descriptors are not what extend_w64.hip needs (see above); "
                           "rebuild with `python -m scratchpad_amd.build --force`")


def build_native(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(CSRC, "*.inc"))) + [
        os.path.join(ROOT, "include", "scratchpad_hip.h")]
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
    # the extend_w64 kernels own accumulation registers the compiler was never told about: size their allocation
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "patch_w64_descriptor.py"), LIB], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)
    check_w64_descriptors(LIB, verbose)
    with open(stamp, "w") as f:
        f.write(made_of)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv))
