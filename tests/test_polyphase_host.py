"""CPU: the polyphase algebra of ResizeConv3d (dropin/polyphase.py) against its definition, and the HOST plan builder of the
C library (mvsgi_conv3d_up2_poly_plan: no GPU call) against that algebra -- folded main weights in the register-stationary
lane order, face-role tables, edge weights."""
import numpy as np
import pytest

from mvs_gi_amd import _lib
from mvs_gi_amd.dropin import polyphase as P
import polyphase_ref as R


@pytest.mark.parametrize("shape", [(1, 3, 4, 5, 6), (2, 2, 1, 3, 2), (1, 2, 2, 1, 5), (1, 2, 3, 4, 1), (1, 2, 1, 1, 1), (1, 2, 2, 2, 2)])
def test_polyphase_equals_interpolate_then_conv(shape):
    """8 phase convolutions over the low-resolution tensor with per-cell class matrices == conv3d(interpolate(x, x2)) in float64:
    single-cell axes (first AND last), two-cell axes (no interior), odd sizes."""
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape)
    w = rng.standard_normal((4, shape[1], 3, 3, 3))
    assert np.abs(R.reference_up2_conv(x, w) - R.polyphase_up2_conv(x, w)).max() <= 1e-12


def test_class_matrices_depend_on_phase_and_class_only():
    for n in (2, 3, 5, 8):
        for i in range(n):
            for p in (0, 1):
                M, C = P.axis_matrix(p, i, n), P.class_matrix(p, P.cell_class(i, n))
                for t in range(3):
                    if 0 <= i + t - 1 < n:                      # rows that multiply in-range samples
                        assert np.allclose(M[t], C[t])
    # the faces differ from the interior in the centre row only (what the face kernels correct)
    for p in (0, 1):
        for cls in (P.FIRST, P.LAST, P.ONLY):
            d = P.face_delta(p, cls)
            assert np.abs(d[1]).max() > 0 and not d[0].any() and not d[2].any()


def _unpack_rs32(buf, f16=False):
    """inverse of rs32_pack_weights_host: 114,688 bytes -> (hi, lo) [32 cout][32 cin][27] as float (bf16 values, or fp16 with f16)."""
    u = buf.view(np.uint16).reshape(2, 2, 14, 2, 64, 8)         # [slice][cout tile][pair][hi|lo][lane][8]
    hi = np.zeros((32, 32, 27), np.float32)
    lo = np.zeros((32, 32, 27), np.float32)

    def f32(h):
        return h.view(np.float16).astype(np.float32) if f16 else (h.astype(np.uint32) << 16).view(np.float32)
    for sl in range(2):
        for ct in range(2):
            for p in range(14):
                for lane in range(64):
                    kg, co = lane >> 4, ct * 16 + (lane & 15)
                    which = kg & 1
                    k = p if p < 9 else (2 * (p - 9) + which if 2 * (p - 9) + which < 9 else -1)
                    kw = which if p < 9 else 2
                    if k < 0:
                        assert not u[sl, ct, p, :, lane].any()
                        continue
                    ci = sl * 16 + (kg >> 1) * 8
                    hi[co, ci:ci + 8, k * 3 + kw] = f32(u[sl, ct, p, 0, lane])
                    lo[co, ci:ci + 8, k * 3 + kw] = f32(u[sl, ct, p, 1, lane])
    return hi, lo


@pytest.mark.parametrize("dims", [(8, 40, 160), (1, 1, 1), (2, 3, 5)])
def test_host_plan_matches_the_python_algebra(dims):
    lib = _lib.load()
    D, H, W = dims
    rng = np.random.default_rng(7)
    w = (rng.standard_normal((16, 32, 3, 3, 3)) / 30).astype(np.float32)
    n = lib.mvsgi_conv3d_up2_poly_plan_bytes(D, H, W)
    assert n > 0 and lib.mvsgi_conv3d_up2_poly_plan_bytes(0, 1, 1) == 0
    plan = np.zeros(n, np.uint8)
    assert lib.mvsgi_conv3d_up2_poly_plan(w.ctypes.data, plan.ctypes.data, D, H, W) == 0, lib.mvsgi_last_error()
    ints = plan[:32].view(np.int32)
    offs = plan[32:80].view(np.int64)
    assert ints[1:4].tolist() == [D, H, W] and offs[5] == n
    off_main, off_facew, off_roles, off_edgew, off_cells = (int(v) for v in offs[:5])
    # main weight sets: hi + lo of the packed set == the folded fp32 weights to 2^-16
    sets = P.main_weight_sets(w)
    for (pd, cd, ph), want in sets.items():
        o = off_main + ((pd * 4 + cd) * 2 + ph) * 114688
        hi, lo = _unpack_rs32(plan[o:o + 114688])
        got = (hi + lo).reshape(32, 32, 3, 3, 3)
        assert np.abs(got - want).max() <= 2.0 ** -15 * np.abs(want).max()
    # edge weights [group][cell][phase][tap (td, th)][ci][co] == (Md x Mh_class x delta_w) at tw = 1
    n_cells, n_groups = int(ints[5]), int(ints[6])
    ew = plan[off_edgew:off_cells].view(np.float32).reshape(n_groups, n_cells, 8, 9, 32, 16)
    cells = plan[off_cells:off_cells + n_cells * 16].view(np.int32).reshape(n_cells, 4)
    groups = [P.ONLY] if D == 1 else ([P.FIRST, P.LAST] if D == 2 else [P.FIRST, P.INT, P.LAST])
    for e in range(n_cells):
        ch, cw = P.cell_class(int(cells[e, 0]), H), P.cell_class(int(cells[e, 1]), W)
        for gi, cd in enumerate(groups):
            for phase in range(8):
                pd, ph, pw = phase >> 2, (phase >> 1) & 1, phase & 1
                We = P.effective_weights(w, P.class_matrix(pd, cd), P.class_matrix(ph, ch), P.face_delta(pw, cw))
                want = np.transpose(We[:, :, :, :, 1], (2, 3, 1, 0)).reshape(9, 32, 16)
                assert np.abs(ew[gi, e, phase] - want).max() <= 1e-6
    assert lib.mvsgi_conv3d_up2_poly_plan(None, plan.ctypes.data, D, H, W) != 0 and b"null pointer" in lib.mvsgi_last_error()


def test_host_plan_in_the_fp16_split():
    """mvsgi_conv3d_up2_poly_plan_fmt(MVSGI_SPLIT_F16): the folded main weight sets as fp16 pairs -- hi + lo reproduces the folded fp32
    weights to 2^-20 (the bf16 plan: 2^-15) once the weights are in fp16's comfortable range (the caller pre-scales them); fmt 0 is the
    un-suffixed function byte for byte; a bad fmt is refused."""
    lib = _lib.load()
    D, H, W = 2, 3, 5
    rng = np.random.default_rng(9)
    w = (rng.standard_normal((16, 32, 3, 3, 3)) * 200).astype(np.float32)        # pre-scaled: largest weight ~ 2^9..2^10
    n = lib.mvsgi_conv3d_up2_poly_plan_bytes(D, H, W)
    p0, pb, pf = np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    assert lib.mvsgi_conv3d_up2_poly_plan(w.ctypes.data, p0.ctypes.data, D, H, W) == 0
    assert lib.mvsgi_conv3d_up2_poly_plan_fmt(w.ctypes.data, pb.ctypes.data, D, H, W, 0) == 0
    assert lib.mvsgi_conv3d_up2_poly_plan_fmt(w.ctypes.data, pf.ctypes.data, D, H, W, 1) == 0
    assert np.array_equal(p0, pb) and not np.array_equal(p0, pf)
    off_main = int(pf[32:80].view(np.int64)[0])
    for (pd, cd, ph), want in P.main_weight_sets(w).items():
        o = off_main + ((pd * 4 + cd) * 2 + ph) * 114688
        hi, lo = _unpack_rs32(pf[o:o + 114688], f16=True)
        assert np.abs((hi + lo).reshape(32, 32, 3, 3, 3) - want).max() <= 2.0 ** -20 * np.abs(want).max()
    assert lib.mvsgi_conv3d_up2_poly_plan_fmt(w.ctypes.data, pf.ctypes.data, D, H, W, 7) != 0 and b"fmt" in lib.mvsgi_last_error()
