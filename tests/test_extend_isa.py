"""The LDS-DMA extend kernel counts its own memory waits (hipcc does not model the DMA).  Two compiler
behaviours would silently cost the two tiles of prefetch the ring exists for, without failing any
parity test: a `s_waitcnt vmcnt(0)` inside the tile loop (a fence, an alias wait in front of a
ds_read_tr intrinsic, a wait for loads the compiler believes pending) and scratch spills in the loop.
This test compiles the headline instantiation to assembly (hipcc cross-compiles without a GPU) and
checks the loop's text.  DESIGN.md section 4.3."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def dma_kernel_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "extend_mfma.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-honor-nans", "-DSP_EXTEND_ONLY_HEADLINE",
           "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "scratchpad_amd", "csrc"),
           os.path.join(ROOT, "scratchpad_amd", "csrc", "extend_mfma.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    text = open(out).read()
    # bf16, D 128, GK 4, 8 waves, 16-bit pool, PLAIN, DMA (, not persistent)
    m = re.search(r"^(_ZN2sp18extend_mfma_kernelINS_8bf16_tagELi128ELi4ELi8ELb0ELb1ELb1(?:ELb0)?EEEvNS_10ExtendArgsE):[^\n]*\n(.*?)^\.Lfunc_end",
                  text, re.S | re.M)
    assert m, "headline LDS-DMA instantiation not found"
    return m.group(2)


def test_tile_loop_keeps_its_dma_in_flight(dma_kernel_asm):
    lines = [l.strip() for l in dma_kernel_asm.splitlines()]
    barriers = [i for i, l in enumerate(lines) if l.startswith("s_barrier")]
    assert len(barriers) >= 6, "prologue barrier + one per unrolled tile step + the epilogue's"
    loop = lines[barriers[0] + 1:barriers[-2]]            # between the prologue's barrier and the last tile step's (the
                                                          # epilogue has its own barrier behind a full drain)
    assert sum(l.startswith("global_load_lds_dwordx4") for l in loop) >= 16, "4 DMA pieces per tile step"
    drained = [l for l in loop if re.match(r"s_waitcnt.*vmcnt\(0\)", l)]
    assert not drained, f"the tile loop drains the DMA ring: {drained[:3]}"
    counted = [l for l in loop if re.match(r"s_waitcnt vmcnt\(8\)", l)]
    assert len(counted) >= 4, "the hand-counted wait of every tile step survives"
    assert not any("scratch_" in l for l in loop[:len(loop) // 2]), "no spills in the unrolled main loop"
    assert "ds_read_b64_tr_b16" in dma_kernel_asm and "v_mfma_f32_32x32x16_bf16" in dma_kernel_asm
