#!/usr/bin/env python3
"""Generates the instruction schedule of one phase of the register-stationary conv kernel
(mvs_gi_amd/csrc/conv3d_rs.hip): a stream of 336 MFMAs (pair 13 of the previous brick, then pairs 0..12 of the
current one) with every other instruction of the phase -- LDS fragment reads, LDS-DMA staging of the next brick,
accumulator hand-over, the epilogue of the brick two phases back, residual requests, descriptor / address upkeep --
placed BY HAND between individual MFMAs and pinned there with sched_barrier(0).

Why a generator: the MFMAs are inline asm (weights pinned to the accumulator register file), which the compiler's
scheduler treats as opaque; and an in-order wave only hides an instruction behind an MFMA when it sits directly behind
it in program order.  Measured on MI355X (tools/ubench/mfma_filler.hip, one wave per SIMD, cycles per
v_mfma_f32_16x16x32_bf16): bare 16.4; + 1 VALU 16.6; + 2 VALU 17.0; + 3 VALU 21.0; + a ds_read_b128 every third MFMA
16.4, with 1 VALU 16.7, with 2 VALU 20.9; + a permlane16_swap 28.9.  So a slot carries two single-issue
instructions (a ds_read_b128 counts as one), and a result is never consumed in the slot after its LDS request.
Output: csrc/conv3d_rs_phase_main.inc and csrc/conv3d_rs_phase_drain.inc (committed).
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mvs_gi_amd", "csrc")
PL, ROW = 120, 20
NSLOT = 14 * 24
CAP = 2.0


def read_stmt(qn, T, lo):
    """fragment of pair qn, voxel tile T (0, 1 own rows; 2, 3 the partner's rows)"""
    grp, i = ("0", T) if T < 2 else ("1", T - 2)
    if qn < 9:
        k = qn
        imm = ((k // 3) * PL + (k % 3) * ROW + i * ROW) * 64
        base = f"rbin[{grp}]"
    else:
        imm = i * ROW * 64
        base = f"rb2[{grp}][{qn - 9}]"
    if lo:
        imm += 30720
    dst = ("xl" if lo else "xh") + f"[{qn & 1}][{T}]"
    return f"RS_F_READ({dst} = *reinterpret_cast<const bf16x8*>(lds + {base} + {imm});)"


OLD_ORDER = os.environ.get("RS_OLD_ORDER") == "1"
OLD_FIN = True      # element-wise reads of an asm "+a" vector gave stale values (hipcc 7.2): read the accumulator whole


def mfma_order(pair):
    """(term, accumulator) order of a pair's 24 MFMAs.  Term-major keeps the three products of an accumulator 8 MFMAs apart.
    Pair 13 finishes the own accumulators (0..3) early and pair 0 starts the partner's (4..7) late, so that every
    accumulator rests >= 6 MFMAs between its last MFMA and the VALU reads that take it out of the accumulator file (an
    MFMA result is not interlocked against VALU reads; 2 MFMAs of distance were measured NOT to be enough)."""
    if OLD_ORDER:
        return [(t, a) for t in range(3) for a in range(8)]
    if pair == 13:
        return [(0, a) for a in range(8)] + [(1, a) for a in range(4)] + [(2, a) for a in range(4)] + \
               [(1, a) for a in range(4, 8)] + [(2, a) for a in range(4, 8)]
    if pair == 0:
        return [(0, a) for a in range(4)] + [(1, a) for a in range(4)] + [(0, a) for a in range(4, 8)] + \
               [(1, a) for a in range(4, 8)] + [(2, a) for a in range(8)]
    return [(t, a) for t in range(3) for a in range(8)]


def mfma_stmt(pair, idx, first_unit_pair):
    """idx 0..23 inside a pair; terms: wl*xh, wh*xl, wh*xh"""
    term, a = mfma_order(pair)[idx]
    T, j = a // 2, a % 2
    w = ("wl" if term == 0 else "wh") + f"[{pair}][{j}]"
    x = ("xl" if term == 1 else "xh") + f"[{pair & 1}][{T}]"
    if first_unit_pair and term == 0:
        return f"RS_MF0(acc[{a}], {w}, {x})"
    return f"RS_MF(acc[{a}], {w}, {x})"


def epilogue_items(k):
    """(cost, statement) list for own accumulator k (tile i' = k >> 1, cout tile j = k & 1) of the brick two phases
    back; pt{k} (the partner's half) was requested from the exchange scratch in block 1"""
    T, j = k >> 1, k & 1
    s = []
    for e in range(4):
        s.append((1, f"t{k}[{e}] = keepB[{k}][{e}] + pt{k}[{e}];"))
    # polyphase mode: the face correction (raw, pre-scale) read back from the output tensor joins the sum before scale / shift
    for e in range(4):
        # (`| 0u`: an rvalue -- __builtin_bit_cast of a vector-element LVALUE reads element 0 whatever the index, hipcc 7.2)
        s.append((0, f"RS_F_UP2(t{k}[{e}] = t{k}[{e}] + __builtin_bit_cast(float, rresB[{k}][{e}] | 0u);)"))
    for e in range(4):
        s.append((1, f"t{k}[{e}] = __builtin_fmaf(t{k}[{e}], esc[{j}][{e}], esh[{j}][{e}]);"))
    s.append((2.0, f"RS_F_NUP2(sa{k} = __builtin_amdgcn_permlane16_swap(rresB[{k}][0], rresB[{k}][2], false, false);)"))
    s.append((2.0, f"RS_F_NUP2(sb{k} = __builtin_amdgcn_permlane16_swap(rresB[{k}][1], rresB[{k}][3], false, false);)"))
    for e, (src, sh) in enumerate((("sa", True), ("sa", False), ("sb", True), ("sb", False))):
        wid = "RS_W_LO" if sh else "RS_W_HI"      # the low / high 16-bit half of a dword as fp32, in the kernel's split (csrc/split_fmt.hpp)
        s.append((1, f"RS_F_NUP2(rh{k} = {wid}({src}{k}[0]);)"))
        s.append((1, f"RS_F_NUP2(rl{k} = {wid}({src}{k}[1]);)"))
        s.append((1, f"RS_F_NUP2(rh{k} = rh{k} + rl{k};)"))
        s.append((1, f"RS_F_NUP2(t{k}[{e}] = t{k}[{e}] + rh{k};)"))
    for e in range(4):
        s.append((1, f"u{k} = t{k}[{e}] * a.neg_slope;"))
        s.append((1, f"t{k}[{e}] = RS_LRELU_MAX(t{k}[{e}], u{k});"))
        # fp16 split only (cost 0: the bf16 schedule keeps its placement): the range clamp, BEHIND the activation (inside its max a slope of 1 -- or t * slope > 65504 -- passed unclamped)
        s.append((0, f"RS_F_SPL(RS_F_F16(t{k}[{e}] = RS_CLAMP(t{k}[{e}]);))"))
    # fp16 split only: the lane's running maximum |clamped value| for the range report (one v_max3_f32 per two elements; csrc/split_fmt.hpp)
    for p in range(2):
        s.append((1, f"RS_F_SPL(RS_F_F16(satm = sf_sat_acc(satm, t{k}[{2 * p}], t{k}[{2 * p + 1}]);))"))
    for p in range(2):
        s.append((1, f"RS_F_SPL(hb{k}[{p}] = RS_CVT_PK(t{k}[{2 * p}], t{k}[{2 * p + 1}]);)"))
        s.append((1, f"RS_F_SPL(hf{k}[0] = RS_W_LO(hb{k}[{p}]);)"))
        s.append((1, f"RS_F_SPL(hf{k}[1] = RS_W_HI(hb{k}[{p}]);)"))
        s.append((1, f"RS_F_SPL(hf{k}[0] = t{k}[{2 * p}] - hf{k}[0];)"))
        s.append((1, f"RS_F_SPL(hf{k}[1] = t{k}[{2 * p + 1}] - hf{k}[1];)"))
        s.append((1, f"RS_F_SPL(lb{k}[{p}] = RS_CVT_PK(hf{k}[0], hf{k}[1]);)"))
    s.append((2.0, f"RS_F_SPL(sa{k} = __builtin_amdgcn_permlane16_swap(hb{k}[0], lb{k}[0], false, false);)"))
    s.append((2.0, f"RS_F_SPL(sb{k} = __builtin_amdgcn_permlane16_swap(hb{k}[1], lb{k}[1], false, false);)"))
    s.append((0.5, f"RS_F_SPL(outp[{k}] = u32x4{{sa{k}[0], sb{k}[0], sa{k}[1], sb{k}[1]}};)"))
    s.append((0.5, f"RS_F_F32(outp[{k}] = __builtin_bit_cast(u32x4, t{k});)"))
    return [(c, f"RS_F_EPI({st})") for c, st in s]


class Sched:
    def __init__(self):
        self.slots = [[] for _ in range(NSLOT)]
        self.load = [0.0] * NSLOT

    def put(self, slot, cost, stmt):
        self.slots[slot].append(stmt)
        self.load[slot] += cost

    def place(self, start, cost, stmt, end=NSLOT, cap=CAP):
        """first slot >= start with room; returns the slot"""
        s = start
        while s < end and self.load[s] + cost > cap + 1e-9:
            s += 1
        assert s < end, (stmt, start)
        self.put(s, cost, stmt)
        return s


def build(with_pairs: bool):
    S = Sched()
    mf = [[] for _ in range(NSLOT)]
    # A. MFMAs (fixed) and fragment reads (fixed: one per three MFMAs, a pair ahead)
    for b in range(14):
        pair = 13 if b == 0 else b - 1
        qn = 0 if b == 0 else b
        if not with_pairs and b >= 1:
            continue
        for pos in range(24):
            mf[b * 24 + pos].append(mfma_stmt(pair, pos, first_unit_pair=(pair == 0)))
        if with_pairs:
            r = 0
            for T in range(4):
                for lo in (0, 1):
                    S.put(b * 24 + 3 * r, 1, read_stmt(qn, T, lo))
                    r += 1
    # B. descriptors / masks of this phase (scalar work + a few VALU), early in block 0
    S.put(1, 0.5, "dsc_y = RS_DSC_Y(c2, ph >= 2);")
    S.put(2, 2.0, "RS_VOY()")
    S.put(4, 0.5, "dsc_x = RS_DESC(a.x, nx, ph + 1 < n);")
    # C. keepB <- keepA while pair 13 runs
    for k in range(4):
        for e in range(4):
            S.place(5, 1, f"keepB[{k}][{e}] = keepA[{k}][{e}];", end=19)
    for k in range(4):
        for e in range(4):
            S.place(5, 1, f"rresB[{k}][{e}] = rres[{k}][{e}];", end=24)
    # D. the accumulators leave the accumulator file (see mfma_order): own accumulators 0..3 rest from slot 12 + a on and
    #    are overwritten at slot 24 + a; the partner's 4..7 rest from slot 16 + a on and are overwritten at slot 28 + a
    if not with_pairs:
        # no MFMAs behind pair 13 in the drain phases: the distance to the accumulator reads must be real time
        S.put(23, 0, "RS_HAZARD_WAIT()")
    for a in range(8):
        dst = f"keepA[{a}]" if a < 4 else f"snd[{a - 4}]"
        lo, hi = (12 + a + 6, 24 + a) if a < 4 else (16 + a + 6, 28 + a)
        if not with_pairs:
            lo, hi = max(lo, 24), NSLOT - 1
        if OLD_ORDER:
            lo, hi = 19 + a, 24 + a
        if OLD_FIN:
            S.put(lo, 4, f"{dst} = acc[{a}]; asm volatile(\"\" : \"+v\"({dst}));")
            continue
        for e in range(4):
            S.place(lo, 1, f"{dst}[{e}] = acc[{a}][{e}]; asm volatile(\"\" : \"+v\"({dst}[{e}]));", end=hi + 1, cap=3.0)
    # E. the partner's half goes to the exchange scratch; the partner's half of the brick two phases back is requested
    for a in range(4):
        S.place(28 + a + 2, 2.0, f"*reinterpret_cast<f32x4*>(lds + sp_wr + {a * 1024}) = snd[{a}];")
    for k in range(4):
        S.place(38, 1, f"RS_F_EPI(pt{k} = *reinterpret_cast<const f32x4*>(lds + sp_rd + {k * 1024});)")
    # F. LDS-DMA of the next brick: pieces m = 0..14 in pairs 0..7 (a slot each)
    if with_pairs:
        m = 0
        # from the phase's first block on: the image the pieces land in was released at the barrier (its last fragments were
        # read in the previous phase); one block earlier than before = 3 % less time in the fp32-output variant, 0.5 % overall
        b0 = int(os.environ.get("RS_DMA_FIRST_PAIR", "0"))
        step = int(os.environ.get("RS_DMA_STEP", "12"))
        for m in range(15):
            S.place(b0 * 24 + 10 + step * m, 2.0, f"RS_F_DMA(RS_DMA({m}))")
    # G. epilogue of the brick two phases back: in program order, wherever a slot has room, from pair 1 on
    s = 2 * 24
    last = s
    for k in range(4):
        for cost, st in epilogue_items(k):
            s = S.place(s, cost, st, end=12 * 24)
            if cost >= 2.0:
                s += 1                      # nothing behind a swap / store in its own slot
            last = s
    # H. residual of the previous brick, consumed by the NEXT phase's epilogue (through rresB): requested early, so that it
    #    has almost two phases to arrive; the output stores of this phase's epilogue go last and are the 4 youngest
    #    vector-memory operations at the phase's end (s_waitcnt vmcnt(4) then covers exactly the staging of the next brick)
    S.put(25, 0.5, "dsc_r = RS_DSC_R(c1, (int)(ph >= 1) & (int)(ph - 1 < n));")
    S.put(27, 0, "RS_VOC()")        # polyphase mode: this lane's correction offsets for brick c1 (cells on an H / W face), else nothing
    s = 32
    for k in range(4):
        s = S.place(s, 2.0, f"RS_F_RES(rres[{k}], dsc_r, RS_RES_OFF({k >> 1}) + {(k & 1) * 64})") + 2
    s = max(11 * 24, last + 1)
    for k in range(4):
        s = S.place(s, 2.0, f"RS_F_EPI(RS_F_STORE(outp[{k}], dsc_y, voy[{k >> 1}] + {(k & 1) * 64}))") + 3
    # I. the read bases move to the other image once their last reads of this phase are out; scratch halves swap
    tg = [(10 * 24 + 1, "rbin[0] ^= BUF1;"), (10 * 24 + 2, "rbin[1] ^= BUF1;")]
    for q in range(4):
        tg += [((10 + q) * 24 + 4, f"rb2[0][{q}] ^= BUF1;"), ((10 + q) * 24 + 5, f"rb2[1][{q}] ^= BUF1;")]
    tg += [(13 * 24 + 22, "rb2[0][4] ^= BUF1;"), (13 * 24 + 23, "rb2[1][4] ^= BUF1;")]
    if not with_pairs:
        tg = []
    tg += [(11 * 24 + 13, "sp_rd ^= 16384;"), (11 * 24 + 14, "sp_wr ^= 16384;")]
    for sl, st in tg:
        S.place(sl, 1, st)
    # J. the walk moves on (scalar)
    S.put(13 * 24 + 10, 0.5, "c2 = c1; c1 = c0; c0 = nx;")
    S.put(13 * 24 + 13, 0.5, "RS_STEP(nx, c0)")
    out = []
    for i in range(NSLOT):
        if not mf[i] and not S.slots[i]:
            continue
        out.append(f"    // slot {i} (block {i // 24}, pos {i % 24}; filler load {S.load[i]:g})")
        for st in mf[i] + S.slots[i]:
            out.append("    " + st)
        out.append("    __builtin_amdgcn_sched_barrier(0);")
    over = sum(1 for l in S.load if l > CAP)
    return "\n".join(out) + "\n", max(S.load), over, last


def main():
    hdr = "// GENERATED by tools/gen_rs_schedule.py -- do not edit; edit the generator and re-run it.\n"
    for name, wp in (("main", True), ("drain", False)):
        txt, mx, over, last = build(wp)
        open(os.path.join(OUT, f"conv3d_rs_phase_{name}.inc"), "w").write(hdr + txt)
        print(f"{name}: max slot load {mx:g}, {over} overloaded slots, epilogue ends at slot {last} (block {last // 24})")


if __name__ == "__main__":
    main()
