"""Op-level parity on a real MI355X: every kernel is called through the C ABI (ctypes) and checked
against plain numpy/float64 arithmetic on the same (operand-rounded) inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

L = None


@pytest.fixture(scope="module", autouse=True)
def _lib():
    global L
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from outfitx_amd import _lib as lib
    lib.load()
    L = lib
    yield


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def stream():
    return torch.cuda.current_stream().cuda_stream


def to_op(a, dt):
    t = torch.from_numpy(a).cuda()
    return t.to(torch.bfloat16 if dt == "bf16" else torch.float16).contiguous()


DT = {"bf16": 1, "f16": 2}


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (127, 128, 128), (128, 256, 768), (300, 384, 3072), (1000, 768, 768)])
def test_gemm_plain(dt, M, N, K):
    g = np.random.default_rng(M * 7 + N + K)
    A = to_op(g.standard_normal((M, K), dtype=np.float32), dt)
    W = to_op(g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K), dt)
    out = torch.full((M, N), float("nan"), device="cuda")
    L.check(L.load().ofx_gemm(A.data_ptr(), W.data_ptr(), out.data_ptr(), None, None, M, N, K, K, N, 0, 0, 0, DT[dt], stream()))
    want = A.double().cpu().numpy() @ W.double().cpu().numpy().T
    assert rel_err(out.cpu().numpy(), want) < 2e-5


def _split_w(Wf, dt):
    """fp32 [N, K] -> ([N, 2K] operand-type [hi | lo] via ofx_convert mode 3, float64 value hi + lo)."""
    N, K = Wf.shape
    src = dev(Wf)
    td = torch.bfloat16 if dt == "bf16" else torch.float16
    W2 = torch.empty(N, 2 * K, dtype=td, device="cuda")
    L.check(L.load().ofx_convert(src.data_ptr(), W2.data_ptr(), N, K, 3, DT[dt], stream()))
    hi = src.to(td)
    lo = (src - hi.float()).to(td)
    assert torch.equal(W2[:, :K], hi) and torch.equal(W2[:, K:], lo)
    return W2, (hi.double() + lo.double()).cpu().numpy()


@pytest.mark.parametrize("force", [0, 6])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(1, 256, 64), (255, 256, 128), (300, 512, 768), (1000, 768, 3072), (70000, 768, 192)])
def test_gemm_split_weights(force, dt, M, N, K):
    """C = A (W_hi + W_lo)^T with one copy of A: the dual-weight 256x256 kernel (forced, and chosen by the dispatcher at M = 70000)
    and the 128x128 kernel with a wrapping A index; exact arithmetic on the rounded operands, fp32 accumulation."""
    g = np.random.default_rng(M + 3 * N + K)
    A = to_op(g.standard_normal((M, K), dtype=np.float32), dt)
    W2, Wv = _split_w((g.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))).astype(np.float32), dt)
    bias = dev(g.standard_normal(N, dtype=np.float32))
    res = dev(g.standard_normal((M, N), dtype=np.float32))
    out = torch.full((M, N), float("nan"), device="cuda")
    lib = L.load()
    lib.ofx_tune(2, force)
    try:
        L.check(lib.ofx_gemm_w2(A.data_ptr(), W2.data_ptr(), out.data_ptr(), bias.data_ptr(), res.data_ptr(), M, N, K, K, N, N, 0, 0, DT[dt], stream()))
    finally:
        lib.ofx_tune(2, 0)
    want = A.double().cpu().numpy() @ Wv.T + bias.cpu().numpy() + res.cpu().numpy()
    assert rel_err(out.cpu().numpy(), want) < 2e-5


@pytest.mark.parametrize("K", [64, 128, 192])
def test_gemm_split_weights_persistent_grid_is_bit_identical_to_one_block_per_tile(K):
    """The persistent dual-weight kernel at its shallowest depths (K = 64: two k-steps per tile, both peeled iterations fetch the next
    tile's steps and the first refill targets the previous tile's epilogue stage; K = 128: four; the dispatcher takes multiples of 64) on a
    many-tile shape with a ragged
    tile count: grids of one block per tile (knob 11 = 0), one per CU (-1) and an odd 7 blocks must agree bit for bit, and with
    float64 arithmetic on the rounded operands."""
    M, N = 70000, 768
    g = np.random.default_rng(K)
    A = to_op(g.standard_normal((M, K), dtype=np.float32), "f16")
    W2, Wv = _split_w((g.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))).astype(np.float32), "f16")
    bias = dev(g.standard_normal(N, dtype=np.float32))
    lib = L.load()
    outs = []
    lib.ofx_tune(2, 6)
    try:
        for persist in (0, -1, 7):
            lib.ofx_tune(11, persist)
            out = torch.full((M, N), float("nan"), device="cuda")
            L.check(lib.ofx_gemm_w2(A.data_ptr(), W2.data_ptr(), out.data_ptr(), bias.data_ptr(), None, M, N, K, K, N, 0, 0, 0, DT["f16"], stream()))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib.ofx_tune(11, -1); lib.ofx_tune(2, 0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel_err(outs[0].cpu().numpy(), A.double().cpu().numpy() @ Wv.T + bias.cpu().numpy()) < 2e-5


def test_profile_records_carry_shape_kernel_and_algorithmic_bytes():
    """ofx_profile_records (what bench.py's roofline block is built from): a GEMM launch is recorded with its logical shape, the
    kernel that ran, its products per term and the ALGORITHMIC HBM bytes its own epilogue configuration implies - A once, the weight
    rows as stored, per output element 2 B (operand type), 4 B + 4 B (fp32 with an fp32 residual), and for the fp8-correction kernel
    2 B hi + 1 B lo per weight."""
    import ctypes as C
    lib = L.load()
    g = np.random.default_rng(3)
    M, N, K = 66000, 1024, 384
    A = to_op(g.standard_normal((M, K), dtype=np.float32), "f16")
    Wf = (g.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))).astype(np.float32)
    W = to_op(Wf, "f16")
    W2, _ = _split_w(Wf, "f16")
    W8 = torch.zeros(N, K, dtype=torch.uint8, device="cuda"); sc = torch.zeros(N, dtype=torch.uint8, device="cuda")
    L.check(lib.ofx_pack_lo8(W2.data_ptr(), W8.data_ptr(), sc.data_ptr(), N, K, stream()))
    o16 = torch.empty(M, N, dtype=torch.float16, device="cuda"); o32 = torch.zeros(M, N, device="cuda")
    lib.ofx_profile_enable(1)
    try:
        L.check(lib.ofx_gemm(A.data_ptr(), W.data_ptr(), o16.data_ptr(), None, None, M, N, K, K, N, 0, 0, 1, DT["f16"], stream()))
        L.check(lib.ofx_gemm(A.data_ptr(), W.data_ptr(), o32.data_ptr(), None, o32.data_ptr(), M, N, K, K, N, N, 0, 0, DT["f16"], stream()))
        L.check(lib.ofx_gemm_w2f8(A.data_ptr(), W2.data_ptr(), W8.data_ptr(), sc.data_ptr(), o16.data_ptr(), None, None, M, N, K, K, N, 0, 0, 1, stream()))
        L.check(lib.ofx_gemm_w2(A.data_ptr(), W2.data_ptr(), o16.data_ptr(), None, None, M, N, K, K, N, 0, 0, 1, DT["f16"], stream()))
        torch.cuda.synchronize()
    finally:
        lib.ofx_profile_enable(0)
    recs = (L.ProfRecord * 16)()
    n = lib.ofx_profile_records(recs, 16)
    ms, fl, cnt = (C.c_double * 4)(), (C.c_double * 4)(), (C.c_longlong * 4)()
    L.check(lib.ofx_profile_read(ms, fl, cnt))
    assert n == 4
    a_b, o2 = 2 * M * K, 2 * M * N
    want = [(1, a_b + 2 * N * K + o2), (1, a_b + 2 * N * K + 8 * M * N), (2, a_b + 3 * N * K + o2), (2, a_b + 4 * N * K + o2)]
    for r, (km, b) in zip(recs, want):
        assert (r.cat, r.M, r.N, r.K, r.kmul) == (0, M, N, K, km) and r.bytes == b and r.ms > 0, (r.M, r.N, r.K, r.kmul, r.kind, r.bytes, b)
    assert recs[2].kind == 8 and recs[3].kind == 6           # the fp8-correction kernel, the dual-weight f16 kernel
    assert recs[2].flops == 0.75 * recs[3].flops               # matrix-pipe work in f16-rate equivalents: 1.5 products against 2


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,ep", [(1, 128, 64, 0), (255, 128, 64, 1), (300, 256, 512, 0), (1000, 512, 2048, 1), (16384, 1536, 512, 2), (40000, 640, 192, 1)])
def test_gemm_three_products_from_operand_tiles_loaded_once(dt, M, N, K, ep):
    """gemm_x3_kernel (forced: ofx_tune(15, 2); chosen by the dispatcher from 192 tiles on): activation rows [hi | lo | hi], weight rows
    [hi | hi | lo], C = hi.hi + lo.hi + hi.lo with each operand tile staged once - against float64 arithmetic on the rounded operands,
    and against the K-concatenated path (ofx_tune(15, 0)) on the same buffers (same products, another summation order).
    ep: 0 fp32 output + bias, 1 fp32 residual in place, 2 operand-type output [hi | lo | hi] + bias."""
    g = np.random.default_rng(M + 5 * N + K)
    td = torch.bfloat16 if dt == "bf16" else torch.float16
    Af = torch.from_numpy(g.standard_normal((M, K), dtype=np.float32)).cuda()
    hi = Af.to(td); lo = (Af - hi.float()).to(td)
    A3 = torch.cat([hi, lo, hi], 1).contiguous()
    Wf = dev((g.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))).astype(np.float32))
    W3 = torch.empty(N, 3 * K, dtype=td, device="cuda")
    L.check(L.load().ofx_convert(Wf.data_ptr(), W3.data_ptr(), N, K, 2, DT[dt], stream()))
    whi, wlo = W3[:, :K].double(), W3[:, 2 * K:].double()
    assert torch.equal(W3[:, :K], W3[:, K:2 * K])
    bias = dev(g.standard_normal(N, dtype=np.float32))
    want = (hi.double() @ whi.T + lo.double() @ whi.T + hi.double() @ wlo.T).cpu().numpy()
    lib = L.load()
    outs = []
    x0 = dev(g.standard_normal((M, N), dtype=np.float32))
    for knob in (2, 0):
        lib.ofx_tune(15, knob)
        try:
            if ep == 2:
                out = torch.zeros(M, 3 * N, dtype=td, device="cuda")
                L.check(lib.ofx_gemm_x3(A3.data_ptr(), W3.data_ptr(), out.data_ptr(), bias.data_ptr(), None, M, N, K, 3 * K, 3 * N, 0, 0, 2, DT[dt], stream()))
                o = out.double().cpu().numpy()
                assert np.array_equal(o[:, :N], o[:, 2 * N:])
                got = o[:, :N] + o[:, N:2 * N]; ref = want + bias.double().cpu().numpy(); tol = 3e-5 if dt == "bf16" else 3e-6
            elif ep == 1:
                x = x0.clone()
                L.check(lib.ofx_gemm_x3(A3.data_ptr(), W3.data_ptr(), x.data_ptr(), None, x.data_ptr(), M, N, K, 3 * K, N, N, 0, 0, DT[dt], stream()))
                got = x.double().cpu().numpy(); ref = want + x0.double().cpu().numpy(); tol = 2e-5
            else:
                out = torch.full((M, N), float("nan"), device="cuda")
                L.check(lib.ofx_gemm_x3(A3.data_ptr(), W3.data_ptr(), out.data_ptr(), bias.data_ptr(), None, M, N, K, 3 * K, N, 0, 0, 0, DT[dt], stream()))
                got = out.double().cpu().numpy(); ref = want + bias.double().cpu().numpy(); tol = 2e-5
            torch.cuda.synchronize()
        finally:
            lib.ofx_tune(15, 1)
        assert rel_err(got, ref) < tol, (knob, rel_err(got, ref))
        outs.append(got)
    assert rel_err(outs[0], outs[1]) < 2 * tol


def _e4m3_decode(b):
    """uint8 ndarray (OCP e4m3fn bit patterns) -> float64 values."""
    b = b.astype(np.int64)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    return np.where(s == 1, -v, v)


def _pack_lo8(W2, N, K):
    """ofx_pack_lo8 on the device; returns (W8 bytes tensor, scale bytes tensor, lo values as the kernel will see them [N, K] float64
    in natural k order = e4m3(byte) 2^-sw, per-row sw)."""
    W8 = torch.zeros(N, K, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(N, dtype=torch.uint8, device="cuda")
    L.check(L.load().ofx_pack_lo8(W2.data_ptr(), W8.data_ptr(), sc.data_ptr(), N, K, stream()))
    torch.cuda.synchronize()
    b = W8.cpu().numpy().reshape(N, K // 128, 4, 4, 8)                 # [n][block][g][s][j] <- k = 32 s + 8 g + j
    nat = _e4m3_decode(b).transpose(0, 1, 3, 2, 4).reshape(N, K)       # [n][block][s][g][j] -> natural k
    scb = sc.cpu().numpy().reshape(N // 128, 16, 8)                    # [(n >> 7)][n & 15][(n >> 4) & 7]
    n = np.arange(N)
    sw = 127 - scb[n >> 7, n & 15, (n >> 4) & 7].astype(np.int64)
    return W8, sc, nat * 2.0 ** (-sw[:, None].astype(np.float64)), sw


@pytest.mark.parametrize("N,K", [(256, 128), (768, 3072), (3072, 768)])
def test_pack_lo8_rounds_and_permutes_as_stated(N, K):
    """fp8 copy of the lo halves: per-row power-of-two scale with max |lo| 2^sw in [128, 256), round-to-nearest-even e4m3, 128-blocks
    k-permuted; against torch's float8_e4m3fn conversion of the same scaled values."""
    g = np.random.default_rng(N + K)
    Wf = (g.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))).astype(np.float32)
    Wf[3] = 0.0                                                          # an all-zero row: lo = 0, scale byte 127
    Wf[5, :7] *= 1e-3                                                    # tiny weights: f16 lo halves that are subnormal
    W2, _ = _split_w(Wf, "f16")
    _, _, lo_seen, sw = _pack_lo8(W2, N, K)
    lo = W2[:, K:].float().cpu()
    mx = lo.abs().max(1).values.numpy().astype(np.float64)
    live = mx > 0
    scaled = mx[live] * 2.0 ** sw[live]
    assert (scaled >= 128).all() and (scaled < 256).all() and (sw[~live] == 0).all()
    want = (lo.double() * torch.from_numpy(2.0 ** sw.astype(np.float64))[:, None]).float().to(torch.float8_e4m3fn).double().numpy() * 2.0 ** (-sw[:, None].astype(np.float64))
    assert np.array_equal(lo_seen, want)
    assert np.abs(lo_seen - lo.double().numpy()).max() <= 2.0 ** -4 * mx.max() * 1.01          # <= half an e4m3 step of the row's top binade


@pytest.mark.parametrize("force", [0, 6])
@pytest.mark.parametrize("M,N,K,big", [(1, 256, 128, 0), (255, 256, 128, 1), (300, 512, 768, 0), (1000, 768, 3072, 2), (70000, 768, 256, 0), (66000, 1024, 384, 1)])
def test_gemm_split_weights_fp8_correction(force, M, N, K, big):
    """C = A hi^T + bf8(A) fp8(lo 2^sw)^T 2^-sw (gemm_w2f8_kernel: forced, and chosen by the dispatcher from 256 tiles on; the small
    problems without force run the f16 lo product of the 128x128 path instead and are held to the exact product): exact arithmetic on
    the quantised operands with fp32 accumulation; and against the float64 product with the unquantised lo halves the correction
    leaves a small fraction of the single-product error - ALSO on activation columns in the hundreds and thousands (big = 1 / 2:
    massive channels, as a trained ViT's residual stream has; round 3's e4m3 image saturated at |a| > 112 and corrected those
    columns only in part, this round's e5m2 image follows f16's whole range) and on columns of 1e-4."""
    g = np.random.default_rng(M + 3 * N + K)
    scales = [[1e-4, 0.01, 1.0, 30.0], [0.01, 1.0, 30.0, 200.0], [1e-4, 1.0, 300.0, 3000.0]][big]
    A = to_op(g.standard_normal((M, K), dtype=np.float32) * g.choice(np.float32(scales), size=(1, K)), "f16")
    W2, Wv = _split_w((g.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))).astype(np.float32), "f16")
    W8, sc, lo_seen, _ = _pack_lo8(W2, N, K)
    bias = dev(g.standard_normal(N, dtype=np.float32))
    res = dev(g.standard_normal((M, N), dtype=np.float32))
    out = torch.full((M, N), float("nan"), device="cuda")
    lib = L.load()
    lib.ofx_tune(2, force)
    try:
        L.check(lib.ofx_gemm_w2f8(A.data_ptr(), W2.data_ptr(), W8.data_ptr(), sc.data_ptr(), out.data_ptr(), bias.data_ptr(), res.data_ptr(), M, N, K, K, N, N, 0, 0, stream()))
        torch.cuda.synchronize()
    finally:
        lib.ofx_tune(2, 0)
    Ad = A.double().cpu().numpy()
    assert np.abs(Ad).max() > (100.0 if big else 1.0)
    hi = W2[:, :K].double().cpu().numpy()
    exact = Ad @ Wv.T + bias.cpu().numpy() + res.cpu().numpy()
    fp8_path = force == 6 or ((M + 255) // 256) * (N // 256) >= 256
    if fp8_path:
        A8 = A.float().clamp(-57344.0, 57344.0).to(torch.float8_e5m2).double().cpu().numpy()
        want = Ad @ hi.T + A8 @ lo_seen.T + bias.cpu().numpy() + res.cpu().numpy()
        assert rel_err(out.cpu().numpy(), want) < 2e-5
        single = Ad @ hi.T + bias.cpu().numpy() + res.cpu().numpy()
        e_corr, e_single = rel_err(out.cpu().numpy(), exact), rel_err(single, exact)
        print(f"w2f8 {M}x{N}x{K} (activation columns up to x{scales[-1]:g}): vs exact split product {e_corr:.2e} (single product {e_single:.2e})")
        assert e_corr < 0.25 * e_single + 2e-5
    else:
        assert rel_err(out.cpu().numpy(), exact) < 2e-5


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("M,N,K,ldc", [(1, 256, 128, 256), (255, 512, 256, 512), (1000, 768, 768, 1536), (70001, 2304, 256, 2304)])
def test_gemm_w2f8_f16_output_leaves_from_the_accumulator_layout(M, N, K, ldc, act):
    """gemm_w2f8_kernel with an operand-type (f16) output, bias and quick-GELU: since round 4 the tile is stored straight from the MFMA accumulator layout
    (epilogue_direct: v_permlane16_swap between neighbouring column blocks, 16 rows x 64 contiguous bytes per store instruction, rows past M dropped by the
    buffer resource's bounds) instead of going through LDS.  Against the float64 product on the quantised operands rounded to f16, bit-identical to the LDS
    epilogue (ofx_tune(18, 0)), ragged M, a row stride wider than N (columns beyond N untouched)."""
    g = np.random.default_rng(M + N + K + act)
    A = to_op(g.standard_normal((M, K), dtype=np.float32), "f16")
    W2, Wv = _split_w((g.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))).astype(np.float32), "f16")
    W8, sc, lo_seen, _ = _pack_lo8(W2, N, K)
    bias = dev(g.standard_normal(N, dtype=np.float32))
    lib = L.load()
    outs = []
    lib.ofx_tune(2, 6)
    try:
        for direct in (1, 0, 2):
            lib.ofx_tune(18, direct)
            out = torch.full((M, ldc), -7.0, dtype=torch.float16, device="cuda")
            L.check(lib.ofx_gemm_w2f8(A.data_ptr(), W2.data_ptr(), W8.data_ptr(), sc.data_ptr(), out.data_ptr(), bias.data_ptr(), None, M, N, K, K, ldc, 0, act, 1, stream()))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib.ofx_tune(2, 0); lib.ofx_tune(18, 1)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], outs[2])                                # 2 = whole-line stores (8 rows x 128 B through a DPP row exchange, predicated global stores)
    got = outs[0].float().cpu().numpy()
    assert (got[:, N:] == -7.0).all()                                   # nothing written beyond the N columns of a wider row
    A8 = A.float().clamp(-57344.0, 57344.0).to(torch.float8_e5m2).double().cpu().numpy()
    z = A.double().cpu().numpy() @ W2[:, :K].double().cpu().numpy().T + A8 @ lo_seen.T + bias.double().cpu().numpy()
    want = z / (1.0 + np.exp(-1.702 * z)) if act else z
    assert rel_err(got[:, :N], want) < 1e-3                             # f16 rounding of the output (2^-11) + the fp32 accumulation


@pytest.mark.parametrize("kern", [2, 3, 4, 5])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(1, 256, 64), (255, 256, 128), (257, 512, 768), (1000, 768, 3072), (5000, 256, 192)])
def test_gemm_big_tile_kernels(kern, dt, M, N, K):
    """The 256x256 (knob 2 = 2) and 256x128 (= 3) kernels forced on for shapes the dispatcher would give to the 128^2 one."""
    lib = L.load()
    g = np.random.default_rng(M + N + K)
    A = to_op(g.standard_normal((M, K), dtype=np.float32), dt)
    W = to_op(g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K), dt)
    bias = dev(g.standard_normal(N, dtype=np.float32))
    x = dev(g.standard_normal((M, N), dtype=np.float32)); x0 = x.clone()
    o3 = torch.zeros(M, 3 * N, dtype=A.dtype, device="cuda")
    lib.ofx_tune(2, kern)
    try:
        L.check(lib.ofx_gemm(A.data_ptr(), W.data_ptr(), x.data_ptr(), bias.data_ptr(), x.data_ptr(), M, N, K, K, N, N, 3, 0, DT[dt], stream()))
        L.check(lib.ofx_gemm(A.data_ptr(), W.data_ptr(), o3.data_ptr(), None, None, M, N, K, K, 3 * N, 0, 0, 2, DT[dt], stream()))
        torch.cuda.synchronize()
    finally:
        lib.ofx_tune(2, 0)
    z = A.double().cpu().numpy() @ W.double().cpu().numpy().T
    assert rel_err(x.cpu().numpy(), O.mish(z + bias.double().cpu().numpy()) + x0.double().cpu().numpy()) < 2e-5
    o3 = o3.float().cpu().numpy()
    assert np.array_equal(o3[:, :N], o3[:, 2 * N:])
    assert rel_err(o3[:, :N] + o3[:, N:2 * N], z) < (3e-5 if dt == "bf16" else 1e-6 + 2e-6)


@pytest.mark.parametrize("M,N,K,kind", [(288, 1024, 3072, 0), (544, 3072, 3072, 0), (100, 2048, 3072, 2), (2304, 1024, 6144, 0), (17, 128, 1024, 1)])
def test_gemm_split_k_is_deterministic_and_exact(M, N, K, kind):
    """Under-filled deep-K shapes (the set transformer at small batch) go through split-K: slab partials + a fused
    reduce/epilogue in a fixed order -> same tolerance as the single-pass kernel and bit-identical run to run."""
    lib = L.load()
    g = np.random.default_rng(M + N + K)
    A = to_op(g.standard_normal((M, K), dtype=np.float32), "bf16")
    W = to_op(g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K), "bf16")
    bias = dev(g.standard_normal(N, dtype=np.float32))
    x0 = dev(g.standard_normal((M, N), dtype=np.float32))
    nb = lib.ofx_gemm_splitk_ws(M, N, K)
    assert nb > 0, "shape was expected to be split"
    slab = torch.empty(nb, dtype=torch.uint8, device="cuda")
    outs = []
    for rep in range(2):
        ld = N * (3 if kind == 2 else 1)
        out = x0.clone() if kind == 0 else torch.zeros(M, ld, dtype=torch.bfloat16, device="cuda")
        L.check(lib.ofx_gemm_splitk(A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), out.data_ptr() if kind == 0 else None,
                                    M, N, K, K, ld, N if kind == 0 else 0, 3, kind, 1, slab.data_ptr(), nb, stream()))
        outs.append(out.float().cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    z = O.mish(A.double().cpu().numpy() @ W.double().cpu().numpy().T + bias.double().cpu().numpy())
    if kind == 0:
        assert rel_err(outs[0], z + x0.double().cpu().numpy()) < 2e-5
    elif kind == 1:
        assert rel_err(outs[0], z) < 2 ** -8
    else:
        assert rel_err(outs[0][:, :N] + outs[0][:, N:2 * N], z) < 3e-5


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,kdev,split", [(256, 256, 64, None, False), (512, 256, 200, None, False), (1024, 512, 2304, 2000, True),
                                              (256, 768, 1, None, True), (768, 256, 4352, 2305, True), (256, 256, 640, 0, True)])
def test_gemm_tn_weight_gradient_kernel(dt, M, N, K, kdev, split):
    """C = A^T B with the contraction over ROWS (transposed LDS reads): asymmetric operands, K not a multiple of 64,
    device-side live row count (rows past it are garbage / NaN and must not contribute), with and without split-K."""
    lib = L.load()
    g = np.random.default_rng(M + N + K)
    Kp = (K + 63) // 64 * 64
    lda, ldb = M + 8, N + 16
    a = g.standard_normal((Kp, lda), dtype=np.float32)
    b = g.standard_normal((Kp, ldb), dtype=np.float32) / np.sqrt(max(K, 1))
    live = K if kdev is None else kdev
    a[live:] = np.nan; b[live:] = np.nan
    A, B = to_op(a, dt), to_op(b, dt)
    kd = None if kdev is None else torch.tensor([kdev], dtype=torch.int32, device="cuda")
    nb = lib.ofx_gemm_tn_ws(M, N, K) if split else 0
    slab = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    outs = []
    for rep in range(2):
        out = torch.full((M, N + 4), float("nan"), device="cuda")
        L.check(lib.ofx_gemm_tn(A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), N + 4, M, N, K, None if kd is None else kd.data_ptr(),
                                slab.data_ptr() if nb else None, nb, DT[dt], stream()))
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0][:, :N], outs[1][:, :N]) and np.isnan(outs[0][:, N:]).all()
    want = A[:live, :M].double().cpu().numpy().T @ B[:live, :N].double().cpu().numpy()
    if live == 0:
        assert not outs[0][:, :N].any()
    else:
        assert rel_err(outs[0][:, :N], want) < 2e-5


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_epilogue_bias_act_residual(act):
    M, N, K = 333, 256, 512
    g = np.random.default_rng(act)
    A = to_op(g.standard_normal((M, K), dtype=np.float32), "bf16")
    W = to_op(g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K) * 2, "bf16")
    bias = dev(g.standard_normal(N, dtype=np.float32))
    x = dev(g.standard_normal((M, N), dtype=np.float32))
    x0 = x.clone()
    # in-place fp32 residual: x += act(A W^T + b)
    L.check(L.load().ofx_gemm(A.data_ptr(), W.data_ptr(), x.data_ptr(), bias.data_ptr(), x.data_ptr(), M, N, K, K, N, N, act, 0, 1, stream()))
    z = A.double().cpu().numpy() @ W.double().cpu().numpy().T + bias.double().cpu().numpy()
    f = {0: lambda u: u, 1: O.quick_gelu, 2: O.gelu, 3: O.mish}[act]
    want = f(z) + x0.double().cpu().numpy()
    assert rel_err(x.cpu().numpy(), want) < 2e-5


def test_gemm_out_kinds_and_split3():
    M, N, K = 200, 128, 192
    g = np.random.default_rng(5)
    A = to_op(g.standard_normal((M, K), dtype=np.float32), "bf16")
    W = to_op(g.standard_normal((N, K), dtype=np.float32), "bf16")
    want = A.double().cpu().numpy() @ W.double().cpu().numpy().T
    o1 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(L.load().ofx_gemm(A.data_ptr(), W.data_ptr(), o1.data_ptr(), None, None, M, N, K, K, N, 0, 0, 1, 1, stream()))
    ref_bf = torch.from_numpy(want).float().to(torch.bfloat16).float().numpy()
    assert np.abs(o1.float().cpu().numpy() - ref_bf).max() <= np.abs(ref_bf).max() * 2 ** -7     # at most 1 bf16 ulp
    o3 = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device="cuda")
    L.check(L.load().ofx_gemm(A.data_ptr(), W.data_ptr(), o3.data_ptr(), None, None, M, N, K, K, 3 * N, 0, 0, 2, 1, stream()))
    o3 = o3.float().cpu().numpy()
    assert np.array_equal(o3[:, :N], o3[:, 2 * N:])
    assert rel_err(o3[:, :N] + o3[:, N:2 * N], want) < 3e-5          # hi + lo carries ~16 bits


def test_gemm_x3_concat_is_fp32_grade():
    """[hi|lo|hi] activations x [hi|hi|lo] weights in ONE K-concatenated GEMM ~ fp32 product."""
    M, N, K = 257, 256, 1024
    g = np.random.default_rng(9)
    A = g.standard_normal((M, K), dtype=np.float32); W = (g.standard_normal((N, K), dtype=np.float32) / 32)
    A3 = torch.empty(M, 3 * K, dtype=torch.bfloat16, device="cuda"); W3 = torch.empty(N, 3 * K, dtype=torch.bfloat16, device="cuda")
    lib = L.load()
    Ad, Wd = dev(A), dev(W)
    L.check(lib.ofx_convert(Ad.data_ptr(), A3.data_ptr(), M, K, 1, 1, stream()))
    L.check(lib.ofx_convert(Wd.data_ptr(), W3.data_ptr(), N, K, 2, 1, stream()))
    out = torch.empty(M, N, device="cuda")
    L.check(lib.ofx_gemm(A3.data_ptr(), W3.data_ptr(), out.data_ptr(), None, None, M, N, 3 * K, 3 * K, N, 0, 0, 0, 1, stream()))
    want = A.astype(np.float64) @ W.astype(np.float64).T
    assert rel_err(out.cpu().numpy(), want) < 2e-5
    # and the single-product bf16 GEMM on the same data is ~100x worse (sanity of the claim)
    out1 = torch.empty(M, N, device="cuda")
    L.check(lib.ofx_gemm(A3.data_ptr(), W3.data_ptr(), out1.data_ptr(), None, None, M, N, K, 3 * K, N, 0, 0, 0, 1, stream()))
    assert rel_err(out1.cpu().numpy(), want) > 5e-4


def test_gemm_rejects_bad_shapes():
    lib = L.load()
    a = torch.zeros(128, 64, dtype=torch.bfloat16, device="cuda")
    o = torch.zeros(128, 128, device="cuda")
    assert lib.ofx_gemm(a.data_ptr(), a.data_ptr(), o.data_ptr(), None, None, 128, 100, 64, 64, 128, 0, 0, 0, 1, stream()) == -2
    assert b"multiple of 128" in lib.ofx_last_error()
    assert lib.ofx_gemm(a.data_ptr(), a.data_ptr(), o.data_ptr(), None, None, 128, 128, 48, 64, 128, 0, 0, 0, 1, stream()) == -2
    assert lib.ofx_gemm(a.data_ptr(), a.data_ptr(), o.data_ptr(), None, None, 0, 128, 64, 64, 128, 0, 0, 0, 1, stream()) == -2


@pytest.mark.parametrize("D", [512, 768, 1024])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_layernorm(D, kind):
    rows = 77
    g = np.random.default_rng(D + kind)
    x = (g.standard_normal((rows + 5, D), dtype=np.float32) * 3 + 1.5)
    gamma = (1 + 0.1 * g.standard_normal(D)).astype(np.float32); beta = (0.1 * g.standard_normal(D)).astype(np.float32)
    idx = g.permutation(rows + 5)[:rows].astype(np.int32)
    want = O.layer_norm(x[idx].astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64))
    ld = D * (3 if kind == 2 else 1)
    y = torch.zeros(rows, ld, dtype=torch.float32 if kind == 0 else torch.bfloat16, device="cuda")
    xd, id_, gd, bd = dev(x), dev(idx), dev(gamma), dev(beta)      # keep alive: the launch is asynchronous
    L.check(L.load().ofx_layernorm(xd.data_ptr(), id_.data_ptr(), gd.data_ptr(), bd.data_ptr(), y.data_ptr(),
                                   rows, D, ld, kind, 1, 1e-5, stream()))
    y = y.float().cpu().numpy()
    if kind == 0:
        assert rel_err(y, want) < 2e-6
    elif kind == 1:
        assert rel_err(y, want) < 2 ** -8
    else:
        assert np.array_equal(y[:, :D], y[:, 2 * D:])
        assert rel_err(y[:, :D] + y[:, D:2 * D], want) < 2e-5


def _attn_ref(q, k, v, scale, dead):
    s = np.einsum("nhqd,nhkd->nhqk", q, k) * scale
    a = O.softmax_masked(s, dead)
    return np.einsum("nhqk,nhkd->nhqd", a, v)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("S,H,causal,masked", [(50, 12, 0, 0), (1, 2, 0, 0), (8, 8, 1, 1), (64, 8, 1, 1), (33, 3, 1, 0), (17, 5, 0, 1)])
def test_attention_mfma(dt, S, H, causal, masked):
    n = 5
    g = np.random.default_rng(S * 31 + H)
    W = H * 64
    qkv = to_op(g.standard_normal((n * S, 3 * W), dtype=np.float32), dt)
    att = np.ones((n, 77), np.int64)
    if masked:
        for i in range(n):
            att[i, g.integers(1, S + 1):] = 0
    out = torch.zeros(n * S, W, dtype=qkv.dtype, device="cuda")
    attd = dev(att)
    L.check(L.load().ofx_attention(qkv.data_ptr(), out.data_ptr(), attd.data_ptr() if masked else None, n, S, H, 3 * W, W, W, 2 * W,
                                   77, causal, 0.125, DT[dt], stream()))
    x = qkv.double().cpu().numpy().reshape(n, S, 3, H, 64).transpose(2, 0, 3, 1, 4)
    dead = np.zeros((n, 1, S, S), bool)
    if masked:
        dead |= (att[:, None, None, :S] == 0)
    if causal:
        dead |= np.triu(np.ones((S, S), bool), 1)[None, None]
    want = _attn_ref(x[0], x[1], x[2], 0.125, dead).transpose(0, 2, 1, 3).reshape(n * S, W)
    # P is rounded to the operand type before P.V (as any MFMA attention does) -> ~2^-9 (bf16) / 2^-12 (f16)
    assert rel_err(out.double().cpu().numpy(), want) < (6e-3 if dt == "bf16" else 8e-4)


@pytest.mark.parametrize("S,causal,masked", [(8, 1, 1), (20, 1, 0), (33, 1, 1), (64, 1, 1), (64, 0, 1), (50, 0, 0)])
def test_attention_f32_fixed_length_causal_and_key_mask(S, causal, masked):
    """The fp32 set-attention kernel in its fixed-length mode (three-product text tower): HF's causal AND key-padding mask, up to 64
    rows per sequence, [hi | lo | hi] output - against float64 softmax attention."""
    n, H = 6, 8
    D = H * 64
    g = np.random.default_rng(S + 3 * causal + masked)
    qkv = g.standard_normal((n * S, 3 * D), dtype=np.float32)
    att = np.ones((n, 77), np.int64)
    if masked:
        for i in range(n):
            att[i, g.integers(1, S + 1):] = 0
    out = torch.zeros(n * S, 3 * D, dtype=torch.float16, device="cuda")
    qd, ad = dev(qkv), dev(att)
    L.check(L.load().ofx_attention_f32(qd.data_ptr(), out.data_ptr(), ad.data_ptr() if masked else None, n, S, H, D, 3 * D, 2, 77, causal, 0.125, DT["f16"], stream()))
    x = qkv.astype(np.float64).reshape(n, S, 3, H, 64).transpose(2, 0, 3, 1, 4)
    dead = np.zeros((n, 1, S, S), bool)
    if masked:
        dead |= (att[:, None, None, :S] == 0)
    if causal:
        dead |= np.triu(np.ones((S, S), bool), 1)[None, None]
    want = _attn_ref(x[0], x[1], x[2], 0.125, dead).transpose(0, 2, 1, 3).reshape(n * S, D)
    o = out.double().cpu().numpy()
    assert np.array_equal(o[:, :D], o[:, 2 * D:])                       # [hi | lo | hi]
    assert rel_err(o[:, :D] + o[:, D:2 * D], want) < 2e-6               # hi + lo = 22 bits of an fp32 result


@pytest.mark.parametrize("fold", [0, 1])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("n,S,H", [(1, 50, 12), (5, 50, 12), (13, 50, 12), (7, 64, 4), (9, 33, 2), (8, 41, 2), (13, 43, 2)])
def test_fused_qkv_projection_attention(fold, dt, n, S, H):
    """The fused QKV-projection + attention kernel (q | k | v staged in LDS only) against (a) the unfused pair ofx_gemm ->
    ofx_attention on the same operands - the same arithmetic in the same order, so equal to the rounding of q | k | v - and
    (b) float64 arithmetic on the operand-rounded inputs.  Image counts below, at and above one block's group (5 at S = 50,
    ragged tail), sequence lengths at both ends of the supported range, with and without the LayerNorm-fold epilogue."""
    g = np.random.default_rng(n * 7 + S + H + fold)
    W = H * 64
    rows = n * S
    X = to_op(g.standard_normal((rows, W), dtype=np.float32), dt)
    Wq = to_op((g.standard_normal((3 * W, W), dtype=np.float32) / np.float32(np.sqrt(W))).astype(np.float32), dt)
    bias = dev((0.1 * g.standard_normal(3 * W)).astype(np.float32))
    stat = dev(np.stack([0.05 * g.standard_normal(rows), 1.0 + 0.1 * g.standard_normal(rows)], 1).astype(np.float32)) if fold else None
    cs = dev((0.2 * g.standard_normal(3 * W)).astype(np.float32)) if fold else None
    lib = L.load()
    out = torch.full((rows, W), float("nan"), dtype=X.dtype, device="cuda")
    L.check(lib.ofx_fused_qkv_attention(X.data_ptr(), Wq.data_ptr(), bias.data_ptr(), stat.data_ptr() if fold else None, cs.data_ptr() if fold else None,
                                        out.data_ptr(), n, S, W, H, W, W, 0.125, DT[dt], stream()))
    # float64 reference of the same (operand-rounded) inputs; q | k | v are rounded to the operand type as the kernel stores them
    qkv = X.double().cpu().numpy() @ Wq.double().cpu().numpy().T
    if fold:
        st = stat.double().cpu().numpy()
        qkv = (qkv - cs.double().cpu().numpy()[None] * st[:, :1]) * st[:, 1:2]
    qkv = qkv + bias.double().cpu().numpy()[None]
    qkv_r = torch.from_numpy(qkv).to(X.dtype).double().numpy()
    x = qkv_r.reshape(n, S, 3, H, 64).transpose(2, 0, 3, 1, 4)
    want = _attn_ref(x[0], x[1], x[2], 0.125, np.zeros((n, 1, S, S), bool)).transpose(0, 2, 1, 3).reshape(rows, W)
    got = out.double().cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, want) < (8e-3 if dt == "bf16" else 1e-3)
    if not fold:        # the unfused pair on the same operands
        q2 = torch.empty(rows, 3 * W, dtype=X.dtype, device="cuda")
        L.check(lib.ofx_gemm(X.data_ptr(), Wq.data_ptr(), q2.data_ptr(), bias.data_ptr(), None, rows, 3 * W, W, W, 3 * W, 0, 0, 1, DT[dt], stream()))
        o2 = torch.zeros(rows, W, dtype=X.dtype, device="cuda")
        L.check(lib.ofx_attention(q2.data_ptr(), o2.data_ptr(), None, n, S, H, 3 * W, W, W, 2 * W, 0, 0, 0.125, DT[dt], stream()))
        assert rel_err(got, o2.double().cpu().numpy()) < (4e-3 if dt == "bf16" else 5e-4)


@pytest.mark.parametrize("S", [35, 36, 42])
def test_fused_qkv_attention_rejects_sequence_lengths_that_overrun_its_pad_rows(S):
    """The last image of a block reads keys up to tile row (G - 1) S + 63; 16 zeroed pad rows sit behind row 255, which S = 35, 36
    and 42 would overrun (18 - 24 rows): the launcher refuses them (OFX_ESHAPE) instead of multiplying P = 0 into stale LDS bytes."""
    W = 2 * 64
    X = torch.zeros(4 * S, W, dtype=torch.float16, device="cuda"); Wq = torch.zeros(3 * W, W, dtype=torch.float16, device="cuda")
    b = torch.zeros(3 * W, device="cuda"); out = torch.zeros(4 * S, W, dtype=torch.float16, device="cuda")
    rc = L.load().ofx_fused_qkv_attention(X.data_ptr(), Wq.data_ptr(), b.data_ptr(), None, None, out.data_ptr(), 4, S, W, 2, W, W, 0.125, DT["f16"], stream())
    assert rc != 0 and b"pad rows" in L.load().ofx_last_error()


@pytest.mark.parametrize("fold", [0, 1])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("n,S,H", [(1, 50, 12), (5, 50, 12), (13, 50, 12), (7, 64, 4), (9, 33, 2)])
def test_fused_qkv_projection_attention_split_weights(fold, dt, n, S, H):
    """Dual-weight variant of the fused kernel (the default tower scheme's ViT layers): q | k | v = X . (hi + lo)^T on the
    gemm_w2 main loop, then the same LDS-staged attention.  Against float64 arithmetic on the operand-rounded X and the exact
    hi + lo weights, and against the unfused pair ofx_gemm_w2 -> ofx_attention on the same operands."""
    g = np.random.default_rng(n * 11 + S + H + fold)
    W = H * 64
    rows = n * S
    X = to_op(g.standard_normal((rows, W), dtype=np.float32), dt)
    W2, Wv = _split_w((g.standard_normal((3 * W, W), dtype=np.float32) / np.float32(np.sqrt(W))).astype(np.float32), dt)
    bias = dev((0.1 * g.standard_normal(3 * W)).astype(np.float32))
    stat = dev(np.stack([0.05 * g.standard_normal(rows), 1.0 + 0.1 * g.standard_normal(rows)], 1).astype(np.float32)) if fold else None
    cs = dev((0.2 * g.standard_normal(3 * W)).astype(np.float32)) if fold else None
    lib = L.load()
    out = torch.full((rows, W), float("nan"), dtype=X.dtype, device="cuda")
    L.check(lib.ofx_fused_qkv_attention_w2(X.data_ptr(), W2.data_ptr(), bias.data_ptr(), stat.data_ptr() if fold else None, cs.data_ptr() if fold else None,
                                           out.data_ptr(), n, S, W, H, W, W, 0.125, DT[dt], stream()))
    qkv = X.double().cpu().numpy() @ Wv.T
    if fold:
        st = stat.double().cpu().numpy()
        qkv = (qkv - cs.double().cpu().numpy()[None] * st[:, :1]) * st[:, 1:2]
    qkv = qkv + bias.double().cpu().numpy()[None]
    qkv_r = torch.from_numpy(qkv).to(X.dtype).double().numpy()
    x = qkv_r.reshape(n, S, 3, H, 64).transpose(2, 0, 3, 1, 4)
    want = _attn_ref(x[0], x[1], x[2], 0.125, np.zeros((n, 1, S, S), bool)).transpose(0, 2, 1, 3).reshape(rows, W)
    got = out.double().cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, want) < (8e-3 if dt == "bf16" else 1e-3)
    if not fold:        # the unfused pair on the same operands
        q2 = torch.empty(rows, 3 * W, dtype=X.dtype, device="cuda")
        L.check(lib.ofx_gemm_w2(X.data_ptr(), W2.data_ptr(), q2.data_ptr(), bias.data_ptr(), None, rows, 3 * W, W, W, 3 * W, 0, 0, 1, DT[dt], stream()))
        o2 = torch.zeros(rows, W, dtype=X.dtype, device="cuda")
        L.check(lib.ofx_attention(q2.data_ptr(), o2.data_ptr(), None, n, S, H, 3 * W, W, W, 2 * W, 0, 0, 0.125, DT[dt], stream()))
        # at these sizes ofx_gemm_w2 runs the 128x128 kernel (another fp32 summation order): single operand-type roundings of q | k | v / the output may differ
        assert rel_err(got, o2.double().cpu().numpy()) < (8e-3 if dt == "bf16" else 1e-3)


@pytest.mark.parametrize("kind,row0", [(0, 0), (2, 0), (0, 1), (1, 0)])
def test_set_attention(kind, row0):
    g = np.random.default_rng(kind * 2 + row0)
    lens = np.array([1, 17, 9, 2, 5, 12, 32, 20])
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows, D, H = int(cu[-1]), 1024, 16
    qkv = g.standard_normal((rows, 3 * D), dtype=np.float32)
    ld = D * (3 if kind == 2 else 1)
    out = torch.zeros(rows, ld, dtype=torch.float32 if kind == 0 else torch.bfloat16, device="cuda")
    qd, cd = dev(qkv), dev(cu)
    L.check(L.load().ofx_set_attention(qd.data_ptr(), out.data_ptr(), cd.data_ptr(), len(lens), H, D, ld, kind, 32, row0, 0.125, 1, stream()))
    out = out.float().cpu().numpy()
    for b, n in enumerate(lens):
        x = qkv[cu[b]:cu[b + 1]].astype(np.float64).reshape(n, 3, H, 64).transpose(1, 2, 0, 3)
        want = _attn_ref(x[0][None], x[1][None], x[2][None], 0.125, np.zeros((1, 1, n, n), bool))[0].transpose(1, 0, 2).reshape(n, D)
        got = out[cu[b]:cu[b + 1]]
        nq = 1 if row0 else n
        if kind == 2:
            assert np.array_equal(got[:nq, :D], got[:nq, 2 * D:])
            got = got[:, :D] + got[:, D:2 * D]
        tol = 2e-6 if kind == 0 else (2e-5 if kind == 2 else 2 ** -8)
        assert rel_err(got[:nq, :D], want[:nq]) < tol
