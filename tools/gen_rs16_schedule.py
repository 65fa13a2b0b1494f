#!/usr/bin/env python3
"""Instruction schedule of one phase of the register-stationary 16 -> 16 conv (post_vol; mvs_gi_amd/csrc/conv3d_rs.hip,
conv3d_rs16_kernel): same method as tools/gen_rs_schedule.py (every non-MFMA instruction placed by hand behind an MFMA, two
single-issue instructions per slot, pinned with sched_barrier(0)), different structure:

  * a wave owns ONE h-row of the 4 x 4 x 16 brick and all 4 output planes (accumulators acc[0..3], Cout = 16 = one tile);
  * "plane" reuse: the fragment of (input plane ip, in-plane tap pair p') is multiplied with the kd = 0, 1, 2 weights into
    output planes ip, ip - 1, ip - 2: 30 fragments (60 ds_read_b128) feed 180 MFMAs;
  * fragments are processed in groups of two (ip and 5 - ip: their MFMAs alternate, so no accumulator is touched twice in
    a row), 15 groups per brick, each read two groups ahead into one of three register sets;
  * one s_barrier per brick between groups 12 and 13: the last two groups cover the first LDS reads of the next brick;
  * no accumulator exchange, no residual: the epilogue of brick u (scale / shift, LeakyReLU, fp32 stores) runs right behind
    its last MFMAs, under the first groups of brick u + 1.
Output: csrc/conv3d_rs16_phase_{main,drain}.inc (committed)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mvs_gi_amd", "csrc")
ROW, PL = 24, 144
REGION = 6 * PL * 32
CAP = 2.0


def frag_addr(ip, pp):
    if pp < 3:
        return "b_in", (ip * PL + pp * ROW) * 32
    if pp == 3:
        return "b_2a", ip * PL * 32
    return "b_2b", (ip * PL + 2 * ROW) * 32


def groups():
    """15 groups per brick: (pp, (ipA, ipB)); the LAST group of a brick is (4, (2, 3)) -- see group_mfmas for its order"""
    return [(pp, pair) for pp in range(5) for pair in ((0, 5), (1, 4), (2, 3))]


def frag_mfmas(ip, pp, slot, first_touch):
    """[(acc, stmt)] of one fragment: kd with 0 <= ip - kd <= 3, three terms each (wl*xh, wh*xl, wh*xh)"""
    out = []
    for term in range(3):
        for kd in range(3):
            op = ip - kd
            if not 0 <= op <= 3:
                continue
            w = ("pwl" if term == 0 else "pwh") + f"[{pp}][{kd}]"
            x = ("xl" if term == 1 else "xh") + f"[GS][{slot}]"
            out.append((op, w, x))
    return out


def group_mfmas(g, first_touched):
    """interleave the two fragments' MFMAs; returns list of statements; marks the first MFMA on each accumulator of a brick"""
    pp, (ia, ib) = groups()[g]
    a, b = frag_mfmas(ia, pp, 0, None), frag_mfmas(ib, pp, 1, None)
    if g == 14:
        # the brick's last group: accumulators 0 and 3 first (they are overwritten first by the next brick), then 1 and 2
        seq = [m for m in a + b if m[0] in (0, 3)]
        seq.sort(key=lambda m: (m[1].startswith("pwh"), m[2].startswith("xh") and m[1].startswith("pwh")))
        a03 = [m for m in seq if m[0] == 0]
        b03 = [m for m in seq if m[0] == 3]
        seq = [x for pair in zip(a03, b03) for x in pair]
        rest_a = [m for m in a if m[0] in (1, 2)]
        rest_b = [m for m in b if m[0] in (1, 2)]
        seq += [x for pair in zip(rest_a, rest_b) for x in pair]
    elif g == 1:
        # the brick's second group is the first to touch accumulators 1 and 2: those MFMAs last (more room for the previous
        # brick's accumulators 1, 2 to be read out)
        early = [m for m in a + b if m[0] in (0, 3)]
        late = [m for m in a + b if m[0] in (1, 2)]
        e0, e3 = [m for m in early if m[0] == 0], [m for m in early if m[0] == 3]
        l1, l2 = [m for m in late if m[0] == 1], [m for m in late if m[0] == 2]
        seq = [x for pair in zip(e0, e3) for x in pair] + [x for pair in zip(l1, l2) for x in pair]
    else:
        seq, i, j = [], 0, 0
        while i < len(a) or j < len(b):
            if i < len(a):
                seq.append(a[i]); i += 1
            if j < len(b):
                seq.append(b[j]); j += 1
    stmts = []
    for op, w, x in seq:
        x = x.replace("GS", str(g % 3))
        if op not in first_touched:
            first_touched.add(op)
            stmts.append((op, f"RS_MF0(acc[{op}], {w}, {x})"))
        else:
            stmts.append((op, f"RS_MF(acc[{op}], {w}, {x})"))
    return stmts


def read_stmts(g):
    pp, (ia, ib) = groups()[g]
    out = []
    for slot, ip in ((0, ia), (1, ib)):
        base, imm = frag_addr(ip, pp)
        out.append(f"RS_F_READ(xh[{g % 3}][{slot}] = *reinterpret_cast<const bf16x8*>(lds + {base} + {imm});)")
        out.append(f"RS_F_READ(xl[{g % 3}][{slot}] = *reinterpret_cast<const bf16x8*>(lds + {base} + {imm + REGION});)")
    return out


class Sched:
    def __init__(self, n):
        self.slots = [[] for _ in range(n)]
        self.load = [0.0] * n

    def put(self, s, cost, st):
        self.slots[s].append(st)
        self.load[s] += cost

    def place(self, start, cost, st, end=None, cap=CAP):
        end = len(self.slots) if end is None else end
        s = start
        while s < end and self.load[s] + cost > cap + 1e-9:
            s += 1
        assert s < end, (st, start)
        self.put(s, cost, st)
        return s


def build(with_pairs, split=False):
    # phase stream: groups 13, 14 of the previous brick, then groups 0..12 of the current one
    mf, gstart, last_mfma = [], {}, {}
    touched = {0, 1, 2, 3}                      # groups 13, 14 continue the previous brick's accumulators
    order = [13, 14] + (list(range(13)) if with_pairs else [])
    for k, g in enumerate(order):
        if k == 2:
            touched = set()
        gstart[(k, g)] = len(mf)
        for op, st in group_mfmas(g, touched):
            if k < 2:
                last_mfma[op] = len(mf)
            mf.append(st)
    n = len(mf) if with_pairs else 30 + 150
    mf += [None] * (n - len(mf))
    S = Sched(n)
    # fragment reads: group at stream position k is read during the group at position k - 2 (positions 0, 1 were read
    # by the previous phase at its positions 13, 14)
    if with_pairs:
        for k in range(len(order)):
            tgt = k + 2
            g_t = order[tgt] if tgt < len(order) else (13 if tgt == len(order) else 14)
            s0 = gstart[(k, order[k])]
            size = (gstart[(k + 1, order[k + 1])] if k + 1 < len(order) else n) - s0
            for q, st in enumerate(read_stmts(g_t)):
                S.place(s0 + min(q * max(size // 4, 1), size - 1), 1, st)
    # upkeep of this phase
    S.put(1, 0.5, "dsc_y = RS16_DESC_OUT(c1, (int)(ph >= 1));")
    S.put(2, 2.0, "RS16_VOY()")
    S.put(4, 0.5, "dsc_x = RS16_DESC(nx, (int)(ph + 1 < n));")
    # the accumulators leave the accumulator file: >= 3 MFMAs after their last MFMA, before the next brick's first MFMA on them
    first_next = {}
    for s, st in enumerate(mf[30:] if with_pairs else []):
        for op in range(4):
            if st and f"RS_MF0(acc[{op}]" in st:
                first_next[op] = 30 + s
    if not with_pairs:
        # no MFMAs behind groups 13, 14 in the drain phase: the distance to the accumulator reads must be real time
        S.put(29, 0, "RS_HAZARD_WAIT()")
    for op in (0, 3, 1, 2):
        lo = last_mfma[op] + 6          # an MFMA result is not interlocked against VALU reads: keep a wide margin
        hi = first_next.get(op, n - 1)
        if not with_pairs:
            lo = max(lo, 30)
        S.place(lo, 4, f"fin[{op}] = acc[{op}]; RS_PIN_V(fin[{op}])", end=hi + 1, cap=5.0)
    # The next brick's LDS-DMA goes FIRST: one piece every 12th slot from slot 6 (the image it lands in was released at the
    # barrier), the epilogue fills the slots around it; nothing orders the two (the phase ends with vmcnt(0)).  The kernel has
    # ONE window in flight per CU and a phase is about as long as a loaded HBM round trip, so every slot the requests go out
    # earlier is time the phase's end does not wait: split-padded output 312 us with the DMA behind the epilogue (slot 113 on),
    # 288 from slot 38, 269 from slot 6 every 10th slot, 264 every 12th.
    dma_start, dma_step = int(os.environ.get("RS16_DMA_START", "6")), int(os.environ.get("RS16_DMA_STEP", "12"))
    early = os.environ.get("RS16_DMA_EARLY_F32", "1") == "1"
    if with_pairs and (split or early):
        for m in range(14):
            S.place(dma_start + dma_step * m, 2.0, f"RS16_F_OWN(RS_F_DMA(RS16_DMA({m})))")
    # epilogue: scale / shift, LeakyReLU, fp32 store -- one instruction per statement
    s = 36
    ep_end = s
    for op in range(4):
        items = [(1, f"fin[{op}][{e}] = __builtin_fmaf(fin[{op}][{e}], esc[{e}], esh[{e}]);") for e in range(4)]
        for e in range(4):
            items.append((1, f"u{op} = fin[{op}][{e}] * a.neg_slope;"))
            items.append((1, f"fin[{op}][{e}] = RS_LRELU_MAX(fin[{op}][{e}], u{op});"))
            if split:      # fp16 split only (cost 0: the bf16 schedule keeps its placement): the range clamp, BEHIND the activation (inside its max a slope of 1 -- or t * slope > 65504 -- passed unclamped)
                items.append((0, f"RS16_F_SPL(RS_F_F16(fin[{op}][{e}] = RS_CLAMP(fin[{op}][{e}]);))"))
        # fp16 split only: the lane's running maximum |clamped value| for the range report (one v_max3_f32 per two elements; csrc/split_fmt.hpp)
        for p in range(2 if split else 0):
            items.append((1, f"RS16_F_SPL(RS_F_F16(satm = sf_sat_acc(satm, fin[{op}][{2 * p}], fin[{op}][{2 * p + 1}]);))"))
        # split-padded output (the stride-2 kernel behind post_vol stages pre-split voxels by LDS-DMA): hi | lo, lanes kg and
        # kg ^ 1 trade halves so that a lane stores 16 contiguous bytes of the voxel record
        for p in range(2 if split else 0):
            items.append((1, f"RS16_F_SPL(hb{op}[{p}] = RS_CVT_PK(fin[{op}][{2 * p}], fin[{op}][{2 * p + 1}]);)"))
            items.append((1, f"RS16_F_SPL(hf{op}[0] = RS_W_LO(hb{op}[{p}]);)"))
            items.append((1, f"RS16_F_SPL(hf{op}[1] = RS_W_HI(hb{op}[{p}]);)"))
            items.append((1, f"RS16_F_SPL(hf{op}[0] = fin[{op}][{2 * p}] - hf{op}[0];)"))
            items.append((1, f"RS16_F_SPL(hf{op}[1] = fin[{op}][{2 * p + 1}] - hf{op}[1];)"))
            items.append((1, f"RS16_F_SPL(lb{op}[{p}] = RS_CVT_PK(hf{op}[0], hf{op}[1]);)"))
        if split:
            items.append((2.0, f"RS16_F_SPL(sa{op} = __builtin_amdgcn_permlane16_swap(hb{op}[0], lb{op}[0], false, false);)"))
            items.append((2.0, f"RS16_F_SPL(sb{op} = __builtin_amdgcn_permlane16_swap(hb{op}[1], lb{op}[1], false, false);)"))
            items.append((2.0, f"RS16_F_SPL(RS_F_STORE16((u32x4{{sa{op}[0], sb{op}[0], sa{op}[1], sb{op}[1]}}), dsc_y, voy[{op}]))"))
        else:
            items.append((2.0, f"RS_F_STORE16(fin[{op}], dsc_y, voy[{op}])"))
        for cost, st in items:
            s = S.place(s, cost, f"RS_F_EPI({st})")
            if cost >= 2.0:
                s += 1
        ep_end = s
    # LDS-DMA of the next brick: 14 pieces per wave, after the epilogue's stores (the staging is then the youngest VMEM work)
    if with_pairs and not split and not early:
        s = ep_end + 2
        step = max((n - 40 - s) // 14, 3)
        for m in range(14):
            s = S.place(s, 2.0, f"RS16_F_OWN(RS_F_DMA(RS16_DMA({m})))") + step
    # the read bases move to the other image once this phase's last reads are out; the walk moves on
    if with_pairs:
        for k, b in enumerate(("b_in", "b_2a", "b_2b")):
            S.place(n - 4 + k, 1, f"{b} ^= BUF1;")
    S.put(n - 8, 0.5, "c1 = c0; c0 = nx;")
    S.put(n - 6, 0.5, "RS16_STEP(nx, c0)")
    out = []
    for i in range(n):
        if mf[i] is None and not S.slots[i]:
            continue
        out.append(f"    // slot {i} (filler load {S.load[i]:g})")
        if mf[i]:
            out.append("    " + mf[i])
        for st in S.slots[i]:
            out.append("    " + st)
        out.append("    __builtin_amdgcn_sched_barrier(0);")
    return "\n".join(out) + "\n", n, max(S.load), ep_end


def main():
    hdr = "// GENERATED by tools/gen_rs16_schedule.py -- do not edit; edit the generator and re-run it.\n"
    for split in (False, True):       # fp32 output / split-padded output (conv3d_rs16_kernel<OSPLIT>)
        for name, wp in (("main", True), ("drain", False)):
            txt, n, mx, ep = build(wp, split)
            open(os.path.join(OUT, f"conv3d_rs16{'s' if split else ''}_phase_{name}.inc"), "w").write(hdr + txt)
            print(f"{'split' if split else 'fp32'} {name}: {n} slots, max slot load {mx:g}, epilogue ends at slot {ep}")


if __name__ == "__main__":
    main()
