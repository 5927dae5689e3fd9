#!/usr/bin/env python3
"""Generates dw_asm.inc: the hand-scheduled main loop of k_spconv_dwa (dwa.hip), the fp32 weight gradient of a sparse
convolution  dW[k][ci][co] += sum over the pairs (in, out) of offset k of  X[in][ci] * dY[out][co]  for one 64 x 64
(ci, co) tile, as ONE inline-asm statement with fixed registers and counted waits (gfx950).

Structure (one wave = one workgroup; four per CU, one per SIMD):
  * a wave owns up to FOUR kernel offsets ("slots": 27 offsets = eight groups of 4, 4, 4, 3, ... dealt longest-first on the
    host) and keeps their 64 x 64 accumulators in the 256 AGPRs for the whole kernel, and a subset of the level's 128-row
    chunks (chunk c, c + step, ...).  A partial tile leaves the chip ONCE per wave and offset (plain stores into the workspace;
    k_dwa_fold adds the waves of an offset in ascending order: reproducible), not once per (row chunk, offset) unit as in
    k_spconv_dw_cmp / k_spconv_dw_reg;
  * per (chunk, slot) = "item": the lane's two rows of the chunk have a neighbour or not -> ballot compaction into a
    wave-private pair list in LDS (byte offsets of the X row and of the dY row), padded to a multiple of FOUR pairs (and at
    least DEPTH groups) with pairs that read a zero row: ~3 % padding where the 16-pair groups of the forward kernel pay 10 %;
  * group = 4 pairs: lane (i, kk) loads 16 bytes of X[in_kk] and of dY[out_kk] (channels 4 i ..); sixteen
    v_mfma_f32_16x16x4_f32 (A = component e of the X piece, B = component f of the dY piece) add the 4-pair product to the
    sixteen 16 x 16 sub-tiles (e, f): no transposition, no LDS staging of operands, no barrier;
  * both gathers run DEPTH = 8 groups ahead (4096 MFMA clocks: rows of the few-row wide levels come from the Infinity
    Cache) — in-place refill of the piece a group has just multiplied, s_waitcnt vmcnt(14) at every group — across item
    boundaries: the last eight groups of an item fetch the first eight of the next (its list is built — exposed, ~200 clocks
    per item — before the item's first group);
  * the register pieces rotate with a flat group counter modulo 8: every slot has eight group bodies and is entered at the
    phase the previous item left.
(First version, EXPERIMENTS.md round 5: seven slots in AGPRs + VGPRs, 64-row chunks, four groups ahead — no registers left
to run further ahead, 115 MB of partial tiles.)
"""
import sys

DEPTH = 8                    # groups the gathers run ahead = register pieces per operand = minimum groups of an item
NSLOT = 4
ROWS = 128                   # rows per chunk (two per lane)
LP = ROWS + 4 * DEPTH + 32   # list capacity (pairs): rows + padding + slack for speculative reads
LDS_IN = 0                   # two lists of LP uint32 (X row byte offsets)
LDS_OUT = 2 * LP * 4         # two lists of LP uint64 (dY row byte offsets)
LDS_DUMP = LDS_OUT + 2 * LP * 8
LDS_BYTES = LDS_DUMP + 64 * 8

XB, DB = 0, 4 * DEPTH        # DEPTH 16-byte pieces each
V = dict(ein=64, eout=66, xoff=68, dyaddr=70, lane=72, i16=73, kk4=74, kk8=75, nv0=76, nv1=77, rowoff0=78, rowoff1=79,
         t0=80, t1=81, t2=82, t3=83, dump=84, zoff=86, outoff0=88, outoff1=90, rowid=92, dyb_lo=93, dyb_hi=94, t4=95)
NV_LAST = 97
S = dict(X=36, DY=38, NB=40, ldx4=42, ldy4=43, nout=44, nslots=45, chunk=46, cstep=47, nchunks=48, part=50, phase=52,
         ngc=53, ngn=54, rem=55, lcur=56, lnxt=57, pin=58, pout=59, t=60, t2=61, m=62, cnt=64, valid0=66, t64=68, lds=70,
         nin=71, kcur=72, zoff=74, slot_nb=76, nout_=84, m2=85, d=86, c=87, pm=88, valid1=90, lbin=92, lbout=93)
S_FIRST, S_LAST = 36, 95


def v(n, w=1):
    return f"v{n}" if w == 1 else f"v[{n}:{n + w - 1}]"


def s(n, w=1):
    return f"s{n}" if w == 1 else f"s[{n}:{n + w - 1}]"


def acc_reg(slot, e, f):
    r = 64 * slot + 4 * (4 * e + f)
    return f"a[{r}:{r + 3}]"


def compaction(lst_sel):
    """Pair list of the NEXT item from v[nv0] / v[nv1] (neighbour rows of the lane's two rows of that item's chunk, -1: none)
    and s[valid0] / s[valid1] (the rows exist).  lst_sel: SGPR holding the list index (0 / 1) to fill.  Leaves the item's
    group count in s[ngn].  Branch-free; lanes without a pair write to their dump slot."""
    m, cnt, t, t2 = S['m'], S['cnt'], S['t'], S['t2']
    lbin, lbout = S['lbin'], S['lbout']
    o = [f"s_mul_i32 {s(lbin)}, {s(lst_sel)}, {LP * 4}",
         f"s_add_u32 {s(lbin)}, {s(lbin)}, {s(S['lds'])}",                   # in-list base (LDS_IN = 0)
         f"s_mul_i32 {s(lbout)}, {s(lst_sel)}, {LP * 8}",
         f"s_add_u32 {s(lbout)}, {s(lbout)}, {s(S['lds'])}",
         f"s_add_u32 {s(lbout)}, {s(lbout)}, {LDS_OUT}"]                      # out-list base
    for j, (nv, valid, outoff) in enumerate(((V['nv0'], S['valid0'], V['outoff0']), (V['nv1'], S['valid1'], V['outoff1']))):
        o += [f"v_cmp_le_i32_e64 {s(m, 2)}, 0, {v(nv)}",
              f"s_and_b64 {s(m, 2)}, {s(m, 2)}, {s(valid, 2)}",
              f"v_mbcnt_lo_u32_b32 {v(V['t0'])}, {s(m)}, 0",
              f"v_mbcnt_hi_u32_b32 {v(V['t0'])}, {s(m + 1)}, {v(V['t0'])}"]
        if j:
            o.append(f"v_add_u32 {v(V['t0'])}, {s(cnt)}, {v(V['t0'])}")
        o += [f"v_mul_lo_u32 {v(V['t1'])}, {v(nv)}, {s(S['ldx4'])}",          # X row byte offset
              f"v_lshl_add_u32 {v(V['t2'])}, {v(V['t0'])}, 2, {s(lbin)}",
              f"v_cndmask_b32_e64 {v(V['t2'])}, {v(V['dump'])}, {v(V['t2'])}, {s(m, 2)}",
              f"ds_write_b32 {v(V['t2'])}, {v(V['t1'])}",
              f"v_lshl_add_u32 {v(V['t3'])}, {v(V['t0'])}, 3, {s(lbout)}",
              f"v_cndmask_b32_e64 {v(V['t3'])}, {v(V['dump'])}, {v(V['t3'])}, {s(m, 2)}",
              f"ds_write_b64 {v(V['t3'])}, {v(outoff, 2)}"]
        if j == 0:
            o.append(f"s_bcnt1_i32_b64 {s(cnt)}, {s(m, 2)}")
        else:
            o += [f"s_bcnt1_i32_b64 {s(t)}, {s(m, 2)}", f"s_add_u32 {s(cnt)}, {s(cnt)}, {s(t)}"]
    # padding: up to a multiple of four pairs and at least DEPTH groups (<= 4 * DEPTH entries: one masked write)
    o += [f"s_add_u32 {s(S['ngn'])}, {s(cnt)}, 3",
          f"s_lshr_b32 {s(S['ngn'])}, {s(S['ngn'])}, 2",
          f"s_max_u32 {s(S['ngn'])}, {s(S['ngn'])}, {DEPTH}",
          f"s_lshl_b32 {s(m)}, {s(S['ngn'])}, 2",
          f"s_sub_u32 {s(m)}, {s(m)}, {s(cnt)}",
          f"v_cmp_gt_i32_e64 {s(S['t64'], 2)}, {s(m)}, {v(V['lane'])}",
          f"v_add_u32 {v(V['t0'])}, {s(cnt)}, {v(V['lane'])}",
          f"v_lshl_add_u32 {v(V['t2'])}, {v(V['t0'])}, 2, {s(lbin)}",
          f"v_cndmask_b32_e64 {v(V['t2'])}, {v(V['dump'])}, {v(V['t2'])}, {s(S['t64'], 2)}",
          f"v_mov_b32 {v(V['t1'])}, 0",
          f"ds_write_b32 {v(V['t2'])}, {v(V['t1'])}",
          f"v_lshl_add_u32 {v(V['t3'])}, {v(V['t0'])}, 3, {s(lbout)}",
          f"v_cndmask_b32_e64 {v(V['t3'])}, {v(V['dump'])}, {v(V['t3'])}, {s(S['t64'], 2)}",
          f"ds_write_b64 {v(V['t3'])}, {v(V['zoff'], 2)}"]
    return o


def row_setup(chunk_sreg):
    """Per-lane row quantities of chunk s[chunk_sreg] (rows 128 c + lane and + 64): s[valid0/1] (the row exists), v[rowoff0/1]
    (byte offset into a neighbour-map row, clamped), v[outoff0/1] (byte offset of the dY row, 64 bit)."""
    o = [f"s_lshl_b32 {s(S['t'])}, {s(chunk_sreg)}, 7",
         f"s_sub_u32 {s(S['t2'])}, {s(S['nout'])}, 1"]
    for j, (valid, rowoff, outoff) in enumerate(((S['valid0'], V['rowoff0'], V['outoff0']), (S['valid1'], V['rowoff1'], V['outoff1']))):
        if j:
            o.append(f"s_add_u32 {s(S['t'])}, {s(S['t'])}, 64")
        o += [f"v_add_u32 {v(V['rowid'])}, {s(S['t'])}, {v(V['lane'])}",
              f"v_cmp_gt_i32_e64 {s(valid, 2)}, {s(S['nout'])}, {v(V['rowid'])}",
              f"v_min_i32 {v(V['rowid'])}, {s(S['t2'])}, {v(V['rowid'])}",
              f"v_lshlrev_b32 {v(rowoff)}, 2, {v(V['rowid'])}",
              f"v_mad_u64_u32 {v(outoff, 2)}, vcc, {v(V['rowid'])}, {s(S['ldy4'])}, 0"]
    return o


def body(slot, u):
    """Group body: phase u (pieces XB + 4u, DB + 4u).  16 MFMAs; fetches the pieces of the group DEPTH ahead: this item's list
    while more than DEPTH groups remain (s[rem] counts the current one), else the next item's."""
    L = [f"L_s{slot}_b{u}_%=:",
         f"s_waitcnt vmcnt({2 * (DEPTH - 1)})"]
    xb, db = XB + 4 * u, DB + 4 * u
    t, t2, m, m2 = S['t'], S['t2'], S['pm'], S['m2']
    gaps = {i: [] for i in range(16)}
    gaps[0] += [f"s_cmp_gt_u32 {s(S['rem'])}, {DEPTH}",
                f"s_cselect_b32 {s(t)}, {s(S['pin'])}, {s(S['nin'])}",
                f"s_cselect_b32 {s(t2)}, {s(S['pout'])}, {s(S['nout_'])}",
                f"s_cselect_b32 {s(m)}, 16, 0"]
    gaps[1] += [f"s_cselect_b32 {s(m2)}, 0, 16",
                f"v_add_u32 {v(V['t0'])}, {s(t)}, {v(V['kk4'])}",
                f"ds_read_b32 {v(V['ein'])}, {v(V['t0'])}",
                f"v_add_u32 {v(V['t1'])}, {s(t2)}, {v(V['kk8'])}"]
    gaps[2] += [f"ds_read_b64 {v(V['eout'], 2)}, {v(V['t1'])}",
                f"s_add_u32 {s(S['pin'])}, {s(S['pin'])}, {s(m)}",
                f"s_add_u32 {s(S['nin'])}, {s(S['nin'])}, {s(m2)}",
                f"s_lshl_b32 {s(m)}, {s(m)}, 1"]
    gaps[3] += [f"s_lshl_b32 {s(m2)}, {s(m2)}, 1",
                f"s_add_u32 {s(S['pout'])}, {s(S['pout'])}, {s(m)}",
                f"s_add_u32 {s(S['nout_'])}, {s(S['nout_'])}, {s(m2)}"]
    gaps[6] += ["s_waitcnt lgkmcnt(0)",
                f"v_add_u32 {v(V['xoff'])}, {v(V['ein'])}, {v(V['i16'])}",
                f"v_add_co_u32_e32 {v(V['dyaddr'])}, vcc, {v(V['eout'])}, {v(V['dyb_lo'])}",
                f"v_addc_co_u32_e32 {v(V['dyaddr'] + 1)}, vcc, {v(V['eout'] + 1)}, {v(V['dyb_hi'])}, vcc"]
    gaps[14] += [f"s_sub_u32 {s(S['rem'])}, {s(S['rem'])}, 1",
                 f"s_cmp_eq_u32 {s(S['rem'])}, 0"]
    mi = 0
    for e in range(4):
        for f in range(4):
            acc = acc_reg(slot, e, f)
            L.append(f"v_mfma_f32_16x16x4_f32 {acc}, {v(xb + e)}, {v(db + f)}, {acc}")
            L += gaps[mi]
            mi += 1
    # refill the two pieces this group has just multiplied with those of the group DEPTH ahead
    import os
    abl = int(os.environ.get("DWA_ABL", "0"))       # experiment builds (wrong results): 1 no X refill, 2 no dY refill
    if not (abl & 1):
        L += [f"global_load_dwordx4 {v(xb, 4)}, {v(V['xoff'])}, {s(S['X'], 2)}"]
    if not (abl & 2):
        L += [f"global_load_dwordx4 {v(db, 4)}, {v(V['dyaddr'], 2)}, off"]
    L += [f"s_cbranch_scc1 L_s{slot}_x{u}_%="]
    return L


def item_ahead(slot, dist):
    """(slot index, chunk) of the item `dist` (1 or 2) behind slot `slot` of the current chunk -> s[d], s[c] (chunk clamped to
    nchunks = a chunk without rows)."""
    L = [f"s_mov_b32 {s(S['d'])}, {slot + dist}", f"s_mov_b32 {s(S['c'])}, {s(S['chunk'])}"]
    for _ in range(dist):
        L += [f"s_cmp_ge_u32 {s(S['d'])}, {s(S['nslots'])}",
              f"s_cselect_b32 {s(S['t'])}, {s(S['nslots'])}, 0",
              f"s_cselect_b32 {s(S['t2'])}, {s(S['cstep'])}, 0",
              f"s_sub_u32 {s(S['d'])}, {s(S['d'])}, {s(S['t'])}",
              f"s_add_u32 {s(S['c'])}, {s(S['c'])}, {s(S['t2'])}"]
    L += [f"s_min_u32 {s(S['c'])}, {s(S['c'])}, {s(S['nchunks'])}"]
    return L


def nb_request():
    """Neighbour rows of item (slot s[d], chunk s[c]) -> v[nv0], v[nv1] (raw: masked when the list is built)."""
    L = [f"s_lshl_b32 {s(S['t2'])}, {s(S['c'])}, 7",
         f"s_sub_u32 {s(S['t'])}, {s(S['nout'])}, 1",
         f"v_add_u32 {v(V['t2'])}, {s(S['t2'])}, {v(V['lane'])}",
         f"v_min_i32 {v(V['t2'])}, {s(S['t'])}, {v(V['t2'])}",
         f"v_lshlrev_b32 {v(V['t2'])}, 2, {v(V['t2'])}",
         f"s_add_u32 {s(S['t2'])}, {s(S['t2'])}, 64",
         f"v_add_u32 {v(V['t3'])}, {s(S['t2'])}, {v(V['lane'])}",
         f"v_min_i32 {v(V['t3'])}, {s(S['t'])}, {v(V['t3'])}",
         f"v_lshlrev_b32 {v(V['t3'])}, 2, {v(V['t3'])}"]
    for k in range(NSLOT):
        L += [f"s_cmp_eq_u32 {s(S['d'])}, {k}",
              f"s_cselect_b32 {s(S['kcur'])}, {s(S['slot_nb'] + 2 * k)}, {s(S['kcur'])}",
              f"s_cselect_b32 {s(S['kcur'] + 1)}, {s(S['slot_nb'] + 2 * k + 1)}, {s(S['kcur'] + 1)}"]
    L += [f"global_load_dword {v(V['nv0'])}, {v(V['t2'])}, {s(S['kcur'], 2)}",
          f"global_load_dword {v(V['nv1'])}, {v(V['t3'])}, {s(S['kcur'], 2)}"]
    return L


def slot_code(slot):
    """glue (list of the next item, request for the one after it, pointers) + DEPTH group bodies + exits."""
    L = [f"L_glue{slot}_%=:"]
    # ---- item t + 1: its rows, its list (its neighbour rows were requested one item ago)
    L += item_ahead(slot, 1)
    L += row_setup(S['c'])
    L += [f"s_waitcnt vmcnt({2 * DEPTH})",
          f"s_xor_b32 {s(S['lnxt'])}, {s(S['lcur'])}, 1"]
    L += compaction(S['lnxt'])
    # ---- item t + 2: request its neighbour rows
    L += item_ahead(slot, 2)
    L += nb_request()
    # ---- pointers: this item's own list from its group DEPTH on; the next item's list from its group 0 on
    L += [f"s_mul_i32 {s(S['t'])}, {s(S['lcur'])}, {LP * 4}",
          f"s_add_u32 {s(S['pin'])}, {s(S['t'])}, {s(S['lds'])}",
          f"s_add_u32 {s(S['pin'])}, {s(S['pin'])}, {16 * DEPTH}",
          f"s_mul_i32 {s(S['t'])}, {s(S['lcur'])}, {LP * 8}",
          f"s_add_u32 {s(S['pout'])}, {s(S['t'])}, {s(S['lds'])}",
          f"s_add_u32 {s(S['pout'])}, {s(S['pout'])}, {LDS_OUT + 32 * DEPTH}",
          f"s_mov_b32 {s(S['nin'])}, {s(S['lbin'])}",             # (the bases compaction has just computed for list lnxt)
          f"s_mov_b32 {s(S['nout_'])}, {s(S['lbout'])}",
          f"s_mov_b32 {s(S['rem'])}, {s(S['ngc'])}",
          "s_waitcnt lgkmcnt(0)"]
    # enter at the phase the previous item left
    for u in range(1, DEPTH):
        L += [f"s_cmp_eq_u32 {s(S['phase'])}, {u}", f"s_cbranch_scc1 L_s{slot}_b{u}_%="]
    for u in range(DEPTH):
        L += body(slot, u)
    L += [f"s_branch L_s{slot}_b0_%="]
    # exits: the item's last group ran in phase u
    for u in range(DEPTH):
        L += [f"L_s{slot}_x{u}_%=:", f"s_mov_b32 {s(S['phase'])}, {(u + 1) % DEPTH}", f"s_branch L_next{slot}_%="]
    L += [f"L_next{slot}_%=:",
          f"s_mov_b32 {s(S['lcur'])}, {s(S['lnxt'])}",
          f"s_mov_b32 {s(S['ngc'])}, {s(S['ngn'])}",
          f"s_cmp_lt_u32 {slot + 1}, {s(S['nslots'])}"]
    if slot + 1 < NSLOT:
        L += [f"s_cbranch_scc1 L_glue{slot + 1}_%="]
    L += ["s_branch L_chunk_%="]
    return L


def program():
    L = []
    L += [f"s_mov_b64 {s(S['X'], 2)}, %[x]",
          f"s_mov_b64 {s(S['DY'], 2)}, %[dy]",
          f"s_mov_b64 {s(S['NB'], 2)}, %[nb]",
          f"s_mov_b32 {s(S['ldx4'])}, %[ldx4]",
          f"s_mov_b32 {s(S['ldy4'])}, %[ldy4]",
          f"s_mov_b32 {s(S['nout'])}, %[nout]",
          f"s_mov_b32 {s(S['nslots'])}, %[nslots]",
          f"s_mov_b32 {s(S['chunk'])}, %[chunk0]",
          f"s_mov_b32 {s(S['cstep'])}, %[cstep]",
          f"s_mov_b32 {s(S['nchunks'])}, %[nchunks]",
          f"s_mov_b32 {s(S['lds'])}, %[lds]",
          f"s_mov_b64 {s(S['zoff'], 2)}, %[zoff]",
          f"s_mov_b64 {s(S['kcur'], 2)}, %[nstride]",          # bytes between two offsets' rows of the neighbour map (temporary)
          f"v_mov_b32 {v(V['lane'])}, %[lane]",
          f"v_mov_b32 {v(V['zoff'])}, {s(S['zoff'])}",
          f"v_mov_b32 {v(V['zoff'] + 1)}, {s(S['zoff'] + 1)}",
          f"v_and_b32 {v(V['t0'])}, 15, {v(V['lane'])}",
          f"v_lshlrev_b32 {v(V['i16'])}, 4, {v(V['t0'])}",
          f"v_lshrrev_b32 {v(V['t1'])}, 4, {v(V['lane'])}",
          f"v_lshlrev_b32 {v(V['kk4'])}, 2, {v(V['t1'])}",
          f"v_lshlrev_b32 {v(V['kk8'])}, 3, {v(V['t1'])}",
          f"v_lshlrev_b32 {v(V['dump'])}, 3, {v(V['lane'])}",
          f"s_add_u32 {s(S['t'])}, {s(S['lds'])}, {LDS_DUMP}",
          f"v_add_u32 {v(V['dump'])}, {s(S['t'])}, {v(V['dump'])}",
          # dY base + 16 i (64 bit) for the per-group address
          f"v_mov_b32 {v(V['dyb_hi'])}, {s(S['DY'] + 1)}",
          f"v_add_co_u32_e32 {v(V['dyb_lo'])}, vcc, {s(S['DY'])}, {v(V['i16'])}",
          f"v_addc_co_u32_e32 {v(V['dyb_hi'])}, vcc, 0, {v(V['dyb_hi'])}, vcc"]
    # neighbour-map row pointers of the wave's slots: NB + k_s * nstride
    for k in range(NSLOT):
        L += [f"s_mul_i32 {s(S['t'])}, %[k{k}], {s(S['kcur'])}",
              f"s_mul_hi_u32 {s(S['t2'])}, %[k{k}], {s(S['kcur'])}",
              f"s_mul_i32 {s(S['m'])}, %[k{k}], {s(S['kcur'] + 1)}",
              f"s_add_u32 {s(S['t2'])}, {s(S['t2'])}, {s(S['m'])}",
              f"s_add_u32 {s(S['slot_nb'] + 2 * k)}, {s(S['NB'])}, {s(S['t'])}",
              f"s_addc_u32 {s(S['slot_nb'] + 2 * k + 1)}, {s(S['NB'] + 1)}, {s(S['t2'])}"]
    for r in range(64 * NSLOT):                # zero the accumulators
        L.append(f"v_accvgpr_write_b32 a{r}, 0")
    # ---- first item (chunk, slot 0): rows, neighbour rows, list 0 (exposed); second item's neighbour rows
    L += row_setup(S['chunk'])
    L += [f"global_load_dword {v(V['nv0'])}, {v(V['rowoff0'])}, {s(S['slot_nb'], 2)}",
          f"global_load_dword {v(V['nv1'])}, {v(V['rowoff1'])}, {s(S['slot_nb'], 2)}",
          "s_waitcnt vmcnt(0)",
          f"s_mov_b32 {s(S['lnxt'])}, 0"]
    L += compaction(S['lnxt'])
    L += [f"s_mov_b32 {s(S['lcur'])}, 0",
          f"s_mov_b32 {s(S['ngc'])}, {s(S['ngn'])}",
          f"s_mov_b32 {s(S['phase'])}, 0",
          "s_waitcnt lgkmcnt(0)"]
    L += item_ahead(0, 1)
    L += nb_request()
    # the first DEPTH groups' pieces (exposed once per wave)
    for u in range(DEPTH):
        L += [f"s_add_u32 {s(S['t'])}, {s(S['lds'])}, {16 * u}",
              f"v_add_u32 {v(V['t0'])}, {s(S['t'])}, {v(V['kk4'])}",
              f"ds_read_b32 {v(V['ein'])}, {v(V['t0'])}",
              f"s_add_u32 {s(S['t'])}, {s(S['lds'])}, {LDS_OUT + 32 * u}",
              f"v_add_u32 {v(V['t1'])}, {s(S['t'])}, {v(V['kk8'])}",
              f"ds_read_b64 {v(V['eout'], 2)}, {v(V['t1'])}",
              "s_waitcnt lgkmcnt(0)",
              f"v_add_u32 {v(V['xoff'])}, {v(V['ein'])}, {v(V['i16'])}",
              f"v_add_co_u32_e32 {v(V['dyaddr'])}, vcc, {v(V['eout'])}, {v(V['dyb_lo'])}",
              f"v_addc_co_u32_e32 {v(V['dyaddr'] + 1)}, vcc, {v(V['eout'] + 1)}, {v(V['dyb_hi'])}, vcc",
              f"global_load_dwordx4 {v(XB + 4 * u, 4)}, {v(V['xoff'])}, {s(S['X'], 2)}",
              f"global_load_dwordx4 {v(DB + 4 * u, 4)}, {v(V['dyaddr'], 2)}, off"]
    L += ["s_branch L_glue0_%="]
    # ---- chunk advance (after the wave's last slot)
    L += ["L_chunk_%=:",
          f"s_add_u32 {s(S['chunk'])}, {s(S['chunk'])}, {s(S['cstep'])}",
          f"s_cmp_lt_u32 {s(S['chunk'])}, {s(S['nchunks'])}",
          "s_cbranch_scc1 L_glue0_%=",
          "s_branch L_done_%="]
    for slot in range(NSLOT):
        L += slot_code(slot)
    L += ["L_done_%=:", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 12"]
    # ---- flush: slot s, sub-tile (e, f), register v -> part[((s * 16 + 4 e + f) * 4 + v) * 64 + lane]
    L += [f"s_mov_b64 {s(S['part'], 2)}, %[part]",
          f"v_lshlrev_b32 {v(V['t0'])}, 2, {v(V['lane'])}"]
    for slot in range(NSLOT):
        L += [f"s_cmp_lt_u32 {slot}, {s(S['nslots'])}", "s_cbranch_scc0 L_flushed_%="]
        for r in range(64):
            L.append(f"global_store_dword {v(V['t0'])}, a{64 * slot + r}, {s(S['part'], 2)} offset:{(r * 256) % 4096}")
            if (r + 1) % 16 == 0:      # 13-bit signed immediate: move the base every 16 registers
                L += [f"s_add_u32 {s(S['part'])}, {s(S['part'])}, 4096", f"s_addc_u32 {s(S['part'] + 1)}, {s(S['part'] + 1)}, 0"]
    L += ["L_flushed_%=:", "s_waitcnt vmcnt(0)"]
    return L


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "dw_asm.inc"
    L = program()
    clob = (["memory", "vcc", "scc"] + [f"v{i}" for i in range(0, NV_LAST + 1)] + [f"a{i}" for i in range(64 * NSLOT)] +
            [f"s{i}" for i in range(S_FIRST, S_LAST + 1)])
    with open(out, "w") as f:
        f.write("// GENERATED by gen_dw_asm.py - do not edit\n")
        f.write(f"#define DWA_LDS_BYTES {LDS_BYTES}\n#define DWA_LP {LP}\n#define DWA_LDS_OUT {LDS_OUT}\n#define DWA_SLOTS {NSLOT}\n"
                f"#define DWA_ROWS {ROWS}\n")
        f.write("#define DWA_ASM_TEXT \\\n")
        for ins in L:
            f.write(f'    "{ins}\\n\\t" \\\n')
        f.write('    ""\n')
        f.write("#define DWA_ASM_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
    print(f"{out}: {len(L)} instructions, {sum(1 for i in L if i.startswith('v_mfma'))} MFMAs", file=sys.stderr)


if __name__ == "__main__":
    main()
