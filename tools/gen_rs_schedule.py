#!/usr/bin/env python3
"""Generates the instruction schedule of one phase of the register-stationary conv kernel
(mvs_gi_amd/csrc/conv3d_rs.hip): a stream of 336 MFMAs (pair 13 of the previous brick, then pairs 0..12 of the
current one) with every other instruction of the phase -- LDS fragment reads, LDS-DMA staging of the next brick,
accumulator hand-over, the epilogue of the brick two phases back, residual requests, address toggles -- placed BY
HAND between individual MFMAs and pinned there with sched_barrier(0).

Why a generator: the MFMAs are inline asm (weights pinned to the accumulator register file), which the compiler's
scheduler treats as opaque; and an in-order wave only hides a VALU / LDS / VMEM instruction behind an MFMA when it sits
directly behind it in program order (an MFMA holds the issue port for 8 of its 16 cycles: two single-issue
instructions fit per MFMA).  Output: csrc/conv3d_rs_phase_main.inc and csrc/conv3d_rs_phase_drain.inc (committed).
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mvs_gi_amd", "csrc")
PL, ROW = 120, 20
NSLOT = 14 * 24


def pair_taps(p):
    """pair p -> (k0, kw0), (k1, kw1) with k = kd*3+kh; in-row pairs first (kw 0|1), then the kw = 2 taps two by two"""
    if p < 9:
        return (p, 0), (p, 1)
    q = p - 9
    return (2 * q, 2), ((2 * q + 1, 2) if 2 * q + 1 < 9 else None)


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
    return f"{dst} = *reinterpret_cast<const bf16x8*>(lds + {base} + {imm});"


def mfma_stmt(pair, idx, first_unit_pair):
    """idx 0..23 inside a pair: term-major (wl*xh, wh*xl, wh*xh), then tile, then cout tile"""
    term, a = idx // 8, idx % 8
    T, j = a // 2, a % 2
    w = ("wl" if term == 0 else "wh") + f"[{pair}][{j}]"
    x = ("xl" if term == 1 else "xh") + f"[{pair & 1}][{T}]"
    if first_unit_pair and term == 0:
        return f"RS_MF0(acc[{a}], {w}, {x})"
    return f"RS_MF(acc[{a}], {w}, {x})"


def epilogue_stmts(k):
    """own accumulator k (tile i' = k >> 1, cout tile j = k & 1) of the brick two phases back"""
    T, j = k >> 1, k & 1
    s = []
    s.append(f"pt{k} = *reinterpret_cast<const f32x4*>(lds + sp_rd + {k * 1024});")
    for e in range(4):
        s.append(f"t{k}[{e}] = keepB[{k}][{e}] + pt{k}[{e}];")
    for e in range(4):
        s.append(f"t{k}[{e}] = __builtin_fmaf(t{k}[{e}], esc[{j}][{e}], esh[{j}][{e}]);")
    s.append(f"sa{k} = __builtin_amdgcn_permlane16_swap(rres[{k}][0], rres[{k}][2], false, false);")
    s.append(f"sb{k} = __builtin_amdgcn_permlane16_swap(rres[{k}][1], rres[{k}][3], false, false);")
    # (own hi, own lo) words: sa = channels 0,1 ; sb = channels 2,3
    for e, (src, sh) in enumerate((("sa", True), ("sa", False), ("sb", True), ("sb", False))):
        hi = f"{src}{k}[0] << 16" if sh else f"{src}{k}[0] & 0xffff0000u"
        lo = f"{src}{k}[1] << 16" if sh else f"{src}{k}[1] & 0xffff0000u"
        s.append(f"rh{k} = __builtin_bit_cast(float, {hi});")
        s.append(f"rl{k} = __builtin_bit_cast(float, {lo});")
        s.append(f"rh{k} = rh{k} + rl{k};")
        s.append(f"t{k}[{e}] = t{k}[{e}] + rh{k};")
    for e in range(4):
        s.append(f"u{k} = t{k}[{e}] * a.neg_slope;")
        s.append(f"t{k}[{e}] = __builtin_fmaxf(t{k}[{e}], u{k});")
    for p in range(2):
        s.append(f"hb{k}[{p}] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{{t{k}[{2 * p}], t{k}[{2 * p + 1}]}}, bf16x2));")
        s.append(f"hf{k}[0] = __builtin_bit_cast(float, hb{k}[{p}] << 16);")
        s.append(f"hf{k}[1] = __builtin_bit_cast(float, hb{k}[{p}] & 0xffff0000u);")
        s.append(f"hf{k}[0] = t{k}[{2 * p}] - hf{k}[0];")
        s.append(f"hf{k}[1] = t{k}[{2 * p + 1}] - hf{k}[1];")
        s.append(f"lb{k}[{p}] = __builtin_bit_cast(unsigned, __builtin_convertvector(hf{k}, bf16x2));")
    s.append(f"sa{k} = __builtin_amdgcn_permlane16_swap(hb{k}[0], lb{k}[0], false, false);")
    s.append(f"sb{k} = __builtin_amdgcn_permlane16_swap(hb{k}[1], lb{k}[1], false, false);")
    s.append(f"__builtin_amdgcn_raw_buffer_store_b128(u32x4{{sa{k}[0], sb{k}[0], sa{k}[1], sb{k}[1]}}, dsc_y, voy[{T}] + {j * 64}, 0, 0);")
    return s


def build(with_pairs: bool):
    slots = [[] for _ in range(NSLOT)]
    # A. MFMAs and fragment reads
    for b in range(14):
        pair = 13 if b == 0 else b - 1
        qn = 0 if b == 0 else b
        if not with_pairs and b >= 1:
            continue
        for pos in range(24):
            slots[b * 24 + pos].append(mfma_stmt(pair, pos, first_unit_pair=(pair == 0)))
        if qn <= 13 and (with_pairs or b == 0):
            if b == 0 and not with_pairs:
                continue           # drain: no next brick to read
            r = 0
            for T in range(4):
                for lo in (0, 1):
                    slots[b * 24 + 3 * r].append(read_stmt(qn, T, lo))
                    r += 1
    # B. keepB <- keepA while pair 13 runs, then the accumulators leave the accumulator file
    for k in range(4):
        for e in range(4):
            slots[k * 4 + e].append(f"keepB[{k}][{e}] = keepA[{k}][{e}];")
    for a in range(8):
        dst = f"keepA[{a}]" if a < 4 else f"snd[{a - 4}]"
        slots[19 + a].append(f"{dst} = acc[{a}]; asm volatile(\"\" : \"+v\"({dst}));")
    # C. the partner's half goes to the exchange scratch
    for a in range(4):
        slots[27 + a].append(f"*reinterpret_cast<f32x4*>(lds + sp_wr + {a * 1024}) = snd[{a}];")
    # D. LDS-DMA of the next brick: pieces m = 0..14 in pairs 0..7
    if with_pairs:
        m = 0
        for b in range(1, 9):
            for pos in (10, 22):
                if m < 15:
                    slots[b * 24 + pos].append(f"RS_DMA({m})")
                    m += 1
    dma_slots = {b * 24 + pos for b in range(1, 9) for pos in (10, 22)}
    # E. epilogue of the brick two phases back: one statement per slot, pairs 1..8
    ep = [st for k in range(4) for st in epilogue_stmts(k)]
    s = 2 * 24
    for st in ep:
        while s in dma_slots:
            s += 1
        slots[s].append(st)
        s += 1
    assert s <= 12 * 24, s
    # F. residual of the previous brick (consumed by the next phase's epilogue): the youngest VMEM operations of the phase
    for k, pos in enumerate((1, 7, 13, 19)):
        slots[12 * 24 + pos].append(f"rres[{k}] = __builtin_amdgcn_raw_buffer_load_b128(dsc_r, voy0[{k >> 1}] + {(k & 1) * 64}, 0, 0);")
    # G. the read bases move to the other image once their last reads of this phase are out
    tg = [(10 * 24 + 2, "rbin[0] ^= BUF1;"), (10 * 24 + 5, "rbin[1] ^= BUF1;")]
    for q in range(4):
        tg += [((10 + q) * 24 + 8, f"rb2[0][{q}] ^= BUF1;"), ((10 + q) * 24 + 11, f"rb2[1][{q}] ^= BUF1;")]
    tg += [(13 * 24 + 22, "rb2[0][4] ^= BUF1;"), (13 * 24 + 23, "rb2[1][4] ^= BUF1;")]
    tg += [(11 * 24 + 14, "sp_rd ^= 16384;"), (11 * 24 + 17, "sp_wr ^= 16384;")]
    if with_pairs:
        for sl, st in tg:
            slots[sl].append(st)
    else:
        slots[11 * 24 + 14].append("sp_rd ^= 16384;")
        slots[11 * 24 + 17].append("sp_wr ^= 16384;")
    out = []
    for i, sl in enumerate(slots):
        if not sl:
            continue
        out.append(f"    // slot {i} (block {i // 24}, pos {i % 24})")
        for st in sl:
            out.append("    " + st)
        out.append("    __builtin_amdgcn_sched_barrier(0);")
    return "\n".join(out) + "\n"


def main():
    hdr = "// GENERATED by tools/gen_rs_schedule.py -- do not edit; edit the generator and re-run it.\n"
    open(os.path.join(OUT, "conv3d_rs_phase_main.inc"), "w").write(hdr + build(True))
    open(os.path.join(OUT, "conv3d_rs_phase_drain.inc"), "w").write(hdr + build(False))
    print("wrote conv3d_rs_phase_{main,drain}.inc")


if __name__ == "__main__":
    main()
