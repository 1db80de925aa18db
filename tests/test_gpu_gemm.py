"""sp_gemm_skinny (decode projections at <= 16 rows) against an fp32 matmul of the same 16-bit
operands: fp32 accumulation and a single rounding, so the result may differ from the exact value
by one rounding of the output dtype plus fp32 summation noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def check(got, x, w, dtype):
    ref = x.float().cpu() @ w.float().cpu().T
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (got.float().cpu() - ref).abs()
    assert bool((err <= eps * ref.abs() + 1e-4 * ref.abs().max()).all()), float(err.max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 3, 8, 16])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (100, 64), (16, 32), (4104, 2048), (28672, 512), (40000, 256)])
def test_skinny_gemm_matches_fp32_reference(dtype, M, N, K):
    from scratchpad_amd import _native
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).cuda()
    got = torch.empty(M, N, dtype=dtype, device="cuda")      # the kernel itself, whatever the dispatch rule says
    _native._check(_native.load().sp_gemm_skinny(got.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0),
                                                 w.stride(0), got.stride(0), 0, _native._dt(x), _native._stream()),
                   "sp_gemm_skinny")
    check(got, x, w, dtype)
    got = _native.linear(x, w)
    assert got.shape == (M, N) and got.dtype == dtype
    check(got, x, w, dtype)
    # the library path on the same operands agrees to the same bound
    check(torch.nn.functional.linear(x, w), x, w, dtype)


def test_skinny_gemm_views_fallbacks_and_determinism():
    from scratchpad_amd import _native
    g = torch.Generator().manual_seed(7)
    big = torch.randn(16, 3 * 4096, generator=g).bfloat16().cuda()
    w = (torch.randn(512, 4096, generator=g) * 0.05).bfloat16().cuda()
    x = big[:5, 4096:8192]                                   # row-strided view, 16-byte aligned
    got = _native.linear(x, w)
    check(got, x, w, torch.bfloat16)
    assert torch.equal(got, _native.linear(x, w)), "fixed summation order: bit-identical reruns"
    x17 = torch.randn(17, 4096, generator=g).bfloat16().cuda()      # > 16 rows: library GEMM
    assert torch.equal(_native.linear(x17, w), torch.nn.functional.linear(x17, w))
    xk = torch.randn(4, 48, generator=g).bfloat16().cuda()          # K % 32 != 0: library GEMM
    wk = torch.randn(8, 48, generator=g).bfloat16().cuda()
    assert torch.equal(_native.linear(xk, wk), torch.nn.functional.linear(xk, wk))
    x32 = torch.randn(2, 64, generator=g).cuda()                    # fp32: library GEMM
    w32 = torch.randn(8, 64, generator=g).cuda()
    assert torch.equal(_native.linear(x32, w32), torch.nn.functional.linear(x32, w32))
    lib = __import__("ctypes").CDLL(_native.lib_path())
    assert lib.sp_gemm_skinny(None, None, None, 4, 8, 64, 64, 64, 8, 0, 2, None) == -1      # null pointers
    assert lib.sp_gemm_skinny(1, 1, 1, 17, 8, 64, 64, 64, 8, 0, 2, None) == -2              # unsupported rows
    assert lib.sp_gemm_skinny(1, 1, 1, 4, 8, 64, 64, 64, 8, 2, 2, None) == -1               # unknown epilogue


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 3, 8, 16])
@pytest.mark.parametrize("I,K", [(14336, 4096), (3584, 4096), (104, 64), (8, 32), (2056, 288)])
def test_silu_mul_fused_into_the_skinny_gate_up_projection_is_bit_identical(dtype, M, I, K):
    """sp_gemm_skinny(epilogue = 1): the merged gate|up projection with SiluAndMul (nn/layers/activation.py:21-31) in
    its epilogue - the bits of the plain skinny projection followed by sp_silu_and_mul, and within one output
    rounding of an fp32 evaluation with torch's own roundings of the activation."""
    from scratchpad_amd import _native
    g = torch.Generator().manual_seed(M * 77 + I + K)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(2 * I, K, generator=g) * (2.0 / K ** 0.5)).to(dtype).cuda()
    w[0].zero_()                                               # a gate column that is exactly 0
    two = torch.empty(M, 2 * I, dtype=dtype, device="cuda")
    _native._check(_native.load().sp_gemm_skinny(two.data_ptr(), x.data_ptr(), w.data_ptr(), M, 2 * I, K, x.stride(0),
                                                 w.stride(0), two.stride(0), 0, _native._dt(x), _native._stream()),
                   "sp_gemm_skinny")
    two_step = _native.silu_and_mul(two)
    fused = _native.linear_silu_mul(x, w, any_rows=True)          # the kernel itself, whatever the dispatch threshold says
    assert fused is not None and fused.shape == (M, I) and torch.equal(fused, two_step)
    assert (_native.linear_silu_mul(x, w) is not None) == (M <= _native.SILU_FUSED_MAX_ROWS)
    gu = (x.float().cpu() @ w.float().cpu().T)
    ref = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (fused.float().cpu() - ref).abs()
    assert bool((err <= 4 * eps * ref.abs() + 4 * eps * ref.abs().max()).all()), float(err.max())
    # a row-strided input view, and the shapes the kernel does not take (the caller keeps the two steps)
    wide = torch.zeros(M + 2, K + 64, dtype=dtype, device="cuda")
    wide[:M, :K] = x
    assert torch.equal(_native.linear_silu_mul(wide[:M, :K], w, any_rows=True), two_step)
    assert _native.linear_silu_mul(torch.zeros(17, K, dtype=dtype, device="cuda"), w, any_rows=True) is None
    assert _native.linear_silu_mul(x, w[:2 * I - 8], any_rows=True) is None


def test_small_step_mlp_uses_the_fused_projection_and_matches_the_two_step_form(monkeypatch):
    from scratchpad_amd import _native, distributed as dist_
    from scratchpad_amd.llama import LlamaMLP
    if not dist_.model_parallel_is_initialized():
        dist_.initialize_model_parallel(1)
    torch.manual_seed(3)
    mlp = LlamaMLP(512, 1408, "silu", torch.bfloat16).cuda()
    for p_ in mlp.parameters():
        p_.data.normal_(0.0, 0.05)
    taken = []
    orig = _native.linear_silu_mul
    monkeypatch.setattr(_native, "linear_silu_mul", lambda a, b: taken.append(orig(a, b) is not None) or orig(a, b))
    for rows in (1, 8, 17):
        x = torch.randn(rows, 512, device="cuda").bfloat16()
        got = mlp(x)
        gate_up = torch.empty(rows, 2816, dtype=torch.bfloat16, device="cuda")
        if rows <= 8:                                           # the skinny projection, unfused, then the activation
            _native._check(_native.load().sp_gemm_skinny(gate_up.data_ptr(), x.data_ptr(), mlp.gate_up_proj.weight.data_ptr(),
                                                         rows, 2816, 512, 512, 512, 2816, 0, _native._dt(x), _native._stream()),
                           "sp_gemm_skinny")
        else:
            gate_up, _ = mlp.gate_up_proj(x)
        want, _ = mlp.down_proj(mlp.act_fn(gate_up))
        assert torch.equal(got, want), rows
    assert taken == [True, True, False]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_hands_the_library_a_faster_row_count_and_returns_the_same_rows(dtype):
    """_native.library_rows: for some (shape, M) the product is computed over M' > M rows of the input's own
    storage (hipBLASLt is faster there).  Rows are independent: whatever the spare rows hold - NaN here - must
    not reach the M rows returned; an input without spare rows is run as asked."""
    from scratchpad_amd import _native
    g = torch.Generator().manual_seed(3)
    for M, N, K in [(256, 6144, 4096), (192, 4096, 14336), (128, 4096, 14336), (16, 28672, 4096), (40, 4096, 4096)]:
        Mp = _native.library_rows(M, N, K)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).cuda()
        x = _native.empty_rows(M, K, dtype, "cuda")
        assert _native.extend_rows(x, Mp) is not None
        torch.as_strided(x, (M + _native.ROW_SLACK, K), (K, 1)).fill_(float("nan"))    # the whole allocation
        x.copy_(torch.randn(M, K, generator=g).to(dtype))
        got = _native.linear(x, w)
        assert got.shape == (M, N) and torch.isfinite(got.float()).all()
        check(got, x, w, dtype)
        assert _native.extend_rows(got, M + _native.ROW_SLACK) is not None, "outputs carry spare rows for the next projection"
        tight = x.clone()                                     # an exact allocation: no spare rows behind it
        if Mp > M:
            assert _native.extend_rows(tight, Mp) is None
        check(_native.linear(tight, w), x, w, dtype)


def test_library_rows_auto_calibration_keeps_only_what_measures_faster_here():
    """SP_LIBRARY_ROWS=auto (_native.calibrate_library_rows): every (shape, bucket) is timed on THIS box against
    M + 8 .. M + ROW_SLACK rows; a substitute is kept only with a measured gain, and whatever the table then says,
    linear() returns the first M rows of the M'-row library product bit for bit."""
    from scratchpad_amd import _native
    saved = {k: dict(v) for k, v in _native._LIBRARY_ROWS.items()}
    g = torch.Generator().manual_seed(5)
    try:
        ws = [(torch.randn(6144, 4096, generator=g) * 0.02).to(torch.bfloat16).cuda() for _ in range(4)]
        table = _native.calibrate_library_rows({(6144, 4096): ws}, [192, 256])
        assert _native._LIBRARY_ROWS is table or _native._LIBRARY_ROWS == table
        for (N, K), sub in table.items():
            assert (N, K) == (6144, 4096)
            for M, Mp in sub.items():
                assert M in (192, 256) and M < Mp <= M + _native.ROW_SLACK and Mp % 8 == 0
        print("auto-calibrated rows for qkv_proj (N 6144, K 4096):", table, "; shipped table:",
              _native._LIBRARY_ROWS_TABLE[(6144, 4096)])
        for M in (192, 256):
            x = _native.empty_rows(M, 4096, torch.bfloat16, "cuda")
            torch.as_strided(x, (M + _native.ROW_SLACK, 4096), (4096, 1)).fill_(float("nan"))
            x.copy_((torch.randn(M, 4096, generator=g) * 0.5).to(torch.bfloat16))
            got = _native.linear(x, ws[0])
            Mp = _native.library_rows(M, 6144, 4096)
            xe = _native.extend_rows(x, Mp)
            ref = torch.mm(xe, ws[0].t())[:M]
            assert torch.equal(got, ref) and torch.isfinite(got.float()).all()
    finally:
        _native._LIBRARY_ROWS = saved
