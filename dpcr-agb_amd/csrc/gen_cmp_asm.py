#!/usr/bin/env python3
"""Generates cmp_asm.inc: the hand-scheduled main loop of k_spconv_cma (spconv.hip), the pair-compacted fp32 sparse
convolution for 128-row x 64-column tiles on gfx950, as ONE inline-asm statement with fixed registers and counted waits.

Why asm: one wave per SIMD (35 KB of running sums per wave in LDS) means that nothing but this wave's own instruction
stream covers its waits, and a v_mfma_f32_16x16x4_f32 leaves the matrix pipe after 32 clocks: every instruction that is not
issued in an MFMA's shadow is matrix-pipe idle time.  hipcc keeps the per-step work (pair compaction of a later offset, the
next neighbour loads, 16 weight loads with their addresses) in its own basic blocks and, once that work is moved into a
group by hand (k_spconv_cmpp experiment, EXPERIMENTS.md round 5), splits the live ranges of the gathered rows across the
blocks (v_mov copies behind an s_waitcnt vmcnt(0): a full gather latency exposed per step).  Here every MFMA gap is filled
explicitly.

Program (per wave = per tile; all 64 lanes active, EXEC never changes):
  init    neighbour rows of offsets 0, 1, 2 -> lists of offsets 0 and 1; weights of step 0; rows of the first group
  step    = 64 input channels of one offset; groups of 16 pairs, 64 MFMAs each:
            D[i][j] += W[k][c][col(i)] * X[in_j][c]   (A = weights, B = gathered rows; lane (j, q) keeps the 16 sums
            of pair j for columns 16q..16q+15 in 16 registers = four ds_read_b128 / ds_write_b128 of its LDS row)
          first group of a step (F1 / F2): + the 16 weight loads of the next step; F2 (first step of an offset): + the
          compaction of offset k + 2 into the third list buffer and the neighbour loads of offset k + 3
  VMEM order inside a group: extras first, then the four refills (rows of the NEXT group, one behind each 16-MFMA block):
          the next group always finds "the last four loads are my rows" -> s_waitcnt vmcnt(3) at every block.
Register map: see the R dictionary.  Waits are computed from the issue order by the generator.
"""
import os
import sys

R_ROWS = 128
YS = 68                      # floats per LDS row of sums
LP = R_ROWS + 16             # list pitch (entries)
LDS_YS = 0
LDS_LISTS = (R_ROWS + 1) * YS * 4          # 35088
LDS_DUMP = LDS_LISTS + 3 * LP * 8          # 38544
LDS_SROW = LDS_DUMP + 64 * 8               # 39056
LDS_BYTES = LDS_SROW + R_ROWS * 4          # 39568

# ---- VGPRs
WA, WB, AN, ACC, CIN = 0, 64, 128, 144, 160
V = dict(lane=176, lst8=177, ysq=178, xq=179, woff=180, rowc0=181, rowc1=182, nv0=184, rid0=185, nv1=186, rid1=187,
         dump=183, e=188, woff1=214, woff2=215, woff3=216, addr_cur=190, addr_nxt=191, xaddr=192, xbc=194, xbn=196, xsel=198, t0=200, t1=201, pf=202, addr_prev=203,
         pad=204, i0=206, i1=208, i2=210, i3=212)
NV_LAST = 235
# ---- SGPRs
S = dict(X=36, Wk=38, Wl=40, N=42, Nstep=44, ldx4=46, cout4=47, wk4=48, K3=49, NSB=50, c13=51, k=52, sb=53, koff=54,
         lb_cur=55, lb_nxt=56, lb_nn=57, ng_cur=58, ng_nxt=59, ng_nn=60, grem=61, steps=62, ng_ns=63, valid0=64, valid1=66,
         m=68, cnt=70, t=71, t2=72, c272=73, last_sb=74, nsb=75, entry=76, lb_ns=77, kv=78, t64=80, lastm=82, lds=84,
         gptr=85, xoffc=86, xoffn=87)
S_FIRST, S_LAST = 36, 87


def v(n, w=1):
    return f"v{n}" if w == 1 else f"v[{n}:{n + w - 1}]"


def s(n, w=1):
    return f"s{n}" if w == 1 else f"s[{n}:{n + w - 1}]"


def mfma(wbase, cb, s2, ct, first):
    a = wbase + 16 * cb + 4 * s2 + ct
    b = AN + 4 * cb + s2
    d = ACC + 4 * ct
    c = CIN + 4 * ct if first else d
    return f"v_mfma_f32_16x16x4_f32 {v(d, 4)}, {v(a)}, {v(b)}, {v(c, 4)}"


def compaction(va, vb, lb, ng_out, kv=None):
    """Pairs of one offset -> list at LDS byte address s[lb]; va / vb: (neighbour row, local row) register pairs of the
    lane's two rows; kv: optional SGPR pair (all ones / zero: the offset exists).  Branch-free: lanes without a pair write
    to their dump slot."""
    m, cnt, t, t2 = S['m'], S['cnt'], S['t'], S['t2']
    out = []

    def one(vx, valid, first):
        o = [f"v_cmp_le_i32_e64 {s(m, 2)}, 0, {v(vx)}",
             f"s_and_b64 {s(m, 2)}, {s(m, 2)}, {s(valid, 2)}"]
        if kv is not None:
            o.append(f"s_and_b64 {s(m, 2)}, {s(m, 2)}, {s(kv, 2)}")
        tv = V['t0'] if first else V['t1']
        o += [f"v_mbcnt_lo_u32_b32 {v(tv)}, {s(m)}, 0",
              f"v_mbcnt_hi_u32_b32 {v(tv)}, {s(m + 1)}, {v(tv)}"]
        if not first:
            o.append(f"v_add_u32 {v(tv)}, {s(cnt)}, {v(tv)}")
        o += [f"v_lshl_add_u32 {v(tv)}, {v(tv)}, 3, {s(lb)}",
              f"v_cndmask_b32_e64 {v(tv)}, {v(V['dump'])}, {v(tv)}, {s(m, 2)}",
              f"ds_write_b64 {v(tv)}, {v(vx, 2)}"]
        if first:
            o.append(f"s_bcnt1_i32_b64 {s(cnt)}, {s(m, 2)}")
        else:
            o += [f"s_bcnt1_i32_b64 {s(t)}, {s(m, 2)}", f"s_add_u32 {s(cnt)}, {s(cnt)}, {s(t)}"]
        return o

    out += one(va, S['valid0'], True)
    out += one(vb, S['valid1'], False)
    out += [f"s_add_u32 {s(t)}, {s(cnt)}, 15",
            f"s_and_b32 {s(t)}, {s(t)}, 0xfffffff0",
            f"s_sub_u32 {s(t2)}, {s(t)}, {s(cnt)}",
            f"v_cmp_gt_i32_e64 {s(m, 2)}, {s(t2)}, {v(V['lane'])}",
            f"v_add_u32 {v(V['t0'])}, {s(cnt)}, {v(V['lane'])}",
            f"v_lshl_add_u32 {v(V['t0'])}, {v(V['t0'])}, 3, {s(lb)}",
            f"v_cndmask_b32_e64 {v(V['t0'])}, {v(V['dump'])}, {v(V['t0'])}, {s(m, 2)}",
            f"ds_write_b64 {v(V['t0'])}, {v(V['pad'], 2)}",
            f"s_lshr_b32 {s(ng_out)}, {s(t)}, 4"]
    return out


def nv_loads(va, vb):
    """Neighbour rows of offset s[koff] (pointer s[N]) for the lane's two rows, then the pointer moves on (stays on the last
    offset past the end: those loads are masked when the list is built)."""
    N, Ns, koff, K3, t64 = S['N'], S['Nstep'], S['koff'], S['K3'], S['t64']
    return [f"global_load_dword {v(va)}, {v(V['rowc0'])}, {s(N, 2)}",
            f"global_load_dword {v(vb)}, {v(V['rowc1'])}, {s(N, 2)}",
            f"s_add_u32 {s(koff)}, {s(koff)}, 1",
            f"s_cmp_lt_u32 {s(koff)}, {s(K3)}",
            f"s_cselect_b32 {s(t64)}, {s(Ns)}, 0",
            f"s_cselect_b32 {s(t64 + 1)}, {s(Ns + 1)}, 0",
            f"s_add_u32 {s(N)}, {s(N)}, {s(t64)}",
            f"s_addc_u32 {s(N + 1)}, {s(N + 1)}, {s(t64 + 1)}"]


def w_loads(wbase):
    """16 dwordx4 weight loads (rows 16 cb + s2 of the 64-channel block at s[Wl]) into the buffer at wbase: the row s2 of
    a 16-row block through four per-lane offsets, the block through the scalar base."""
    Wl = S['Wl']
    woffs = [V['woff'], V['woff1'], V['woff2'], V['woff3']]
    out = []
    for cb in range(4):
        for s2 in range(4):
            out.append(f"global_load_dwordx4 {v(wbase + 16 * cb + 4 * s2, 4)}, {v(woffs[s2])}, {s(Wl, 2)}")
        if cb < 3:
            out += [f"s_add_u32 {s(Wl)}, {s(Wl)}, {s(S['c13'])}", f"s_addc_u32 {s(Wl + 1)}, {s(Wl + 1)}, 0"]
    return out


ABL = 0        # experiment variants (wrong results): 1 no row refills, 2 no weight loads, 4 no sums traffic, 8 no compaction
STAMPS = os.environ.get("CMA_STAMPS", "0") == "1"     # diagnostic build only (make timeline): s_memtime sums per segment
ST = dict(prev=88, now=90, top=92, F=94, P=96, nF=98, nP=99)


def stamp(acc, count=None):
    if not STAMPS:
        return []
    o = [f"s_memtime {s(ST['now'], 2)}", "s_waitcnt lgkmcnt(0)",
         f"s_sub_u32 {s(S['t'])}, {s(ST['now'])}, {s(ST['prev'])}", f"s_subb_u32 {s(S['t2'])}, {s(ST['now'] + 1)}, {s(ST['prev'] + 1)}",
         f"s_add_u32 {s(ST[acc])}, {s(ST[acc])}, {s(S['t'])}", f"s_addc_u32 {s(ST[acc] + 1)}, {s(ST[acc] + 1)}, {s(S['t2'])}",
         f"s_mov_b64 {s(ST['prev'], 2)}, {s(ST['now'], 2)}"]
    if count:
        o.append(f"s_add_u32 {s(ST[count])}, {s(ST[count])}, 1")
    return o


def timed_wait(wait, acc32):
    """(diagnostic build) the wait bracketed by two clock reads, its duration added to the 32-bit sum s[acc32]"""
    if not STAMPS:
        return [wait]
    return [f"s_memtime {s(ST['now'], 2)}", "s_waitcnt lgkmcnt(0)", f"s_mov_b32 {s(S['gptr'])}, {s(ST['now'])}", wait,
            f"s_memtime {s(ST['now'], 2)}", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 {s(S['gptr'])}, {s(ST['now'])}, {s(S['gptr'])}",
            f"s_add_u32 {s(acc32)}, {s(acc32)}, {s(S['gptr'])}"]


def is_vmem(ins):
    return ins.startswith("global_load")


ACCS = (144, 220)      # two accumulator sets: group g + 1 multiplies into one while group g's sums leave the other


def group(wcur, wnxt, F, P, pending):
    """One 16-pair group accumulating in set P.  On entry: rows of the group in AN (refills in flight), its sums in CIN, the list
    entry of the FOLLOWING group in v[e], s[lastm] = this is the step's last group; pending: the previous group's sums still
    sit in set 1 - P and leave for LDS (address v[addr_prev]) behind this group's first MFMAs."""
    lines = []
    after = {i: [] for i in range(-1, 64)}      # instructions issued right behind MFMA i (-1: ahead of the first)
    entry, lastm, grem = S['entry'], S['lastm'], S['grem']
    acc, accp = ACCS[P], ACCS[1 - P]
    after[-1] += timed_wait("s_waitcnt vmcnt(3) lgkmcnt(0)", 100)
    # ---- list entry of the following group -> its sums' address, its rows' address
    addr = [f"v_cndmask_b32_e64 {v(V['xsel'])}, {v(V['xbc'])}, {v(V['xbn'])}, {s(lastm, 2)}",
            f"v_cndmask_b32_e64 {v(V['xsel'] + 1)}, {v(V['xbc'] + 1)}, {v(V['xbn'] + 1)}, {s(lastm, 2)}",
            f"v_mad_u32_u24 {v(V['addr_nxt'])}, {v(V['e'] + 1)}, {s(S['c272'])}, {v(V['ysq'])}",
            f"v_mad_u64_u32 {v(V['xaddr'], 2)}, vcc, {v(V['e'])}, {s(S['ldx4'])}, {v(V['xsel'], 2)}",
            f"v_cndmask_b32_e64 {v(V['pf'])}, {v(V['addr_nxt'])}, {v(V['addr_cur'])}, {s(lastm, 2)}"]
    writes = [f"ds_write_b128 {v(V['addr_prev'])}, {v(accp + 4 * ct, 4)}" + (f" offset:{16 * ct}" if ct else "")
              for ct in range(4)] if pending else []
    # sums of the following group (a dummy re-read of the own rows in a step's last group): behind MFMA 3, the last reader of CIN
    pf = [f"ds_read_b128 {v(CIN + 4 * ct, 4)}, {v(V['pf'])} offset:{16 * ct}" for ct in range(4)]
    extras_valu = []      # compaction (VALU / SALU / LDS writes)
    extras_vmem = []      # loads with their pointer arithmetic
    if F == 2:
        # the neighbour rows of offset k + 2 move to a copy (the loads of offset k + 3 go out right away); kv = (k + 2 < K3)
        extras_vmem += [f"v_mov_b32 {v(V['i0'])}, {v(V['nv0'])}", f"v_mov_b32 {v(V['i1'])}, {v(V['nv1'])}"]
        extras_vmem += nv_loads(V['nv0'], V['nv1'])
        extras_valu += [f"s_add_u32 {s(S['t'])}, {s(S['k'])}, 2",
                        f"s_cmp_lt_u32 {s(S['t'])}, {s(S['K3'])}",
                        f"s_cselect_b64 {s(S['kv'], 2)}, -1, 0"]
        extras_valu += compaction(V['i0'], V['i1'], S['lb_nn'], S['ng_nn'], S['kv'])
    if F >= 1 and not (ABL & 2):
        extras_vmem += w_loads(wnxt)
    if ABL & 4:
        writes, pf = [], []
    if ABL & 8:
        extras_valu = []
    # what the NEXT group needs on entry: is it the step's last, the list entry of the group behind it (from the next step's
    # list when the next group is the last), the addresses moved on
    late = [f"s_sub_u32 {s(grem)}, {s(grem)}, 1",
            f"s_cmp_eq_u32 {s(grem)}, 1",
            f"s_cselect_b64 {s(lastm, 2)}, -1, 0",
            f"s_cselect_b32 {s(S['t'])}, {s(S['lb_ns'])}, {s(entry)}",
            f"s_add_u32 {s(entry)}, {s(entry)}, 128",
            f"v_add_u32 {v(V['t0'])}, {s(S['t'])}, {v(V['lst8'])}",
            f"ds_read_b64 {v(V['e'], 2)}, {v(V['t0'])}",
            f"v_mov_b32 {v(V['addr_prev'])}, {v(V['addr_cur'])}",
            f"v_mov_b32 {v(V['addr_cur'])}, {v(V['addr_nxt'])}"]
    # placement: at most PER instructions per MFMA gap.  Block 0: the loads of the step (neighbour rows, weights: ahead of
    # refill 0) with the address math; the previous group's sums leave from gap 1 on (12 wait states behind its last MFMA);
    # the compaction follows in blocks 1 and 2.  Gaps 15 / 31 / 47 / 63 belong to the refills, 50.. to `late`.
    PER = int(os.environ.get("CMA_PER", "3")) if F else int(os.environ.get("CMA_PER0", "2"))
    n0 = min(len(extras_vmem), 2 * PER)
    order = extras_vmem[:n0] + addr + [1] + writes + extras_vmem[n0:] + [3] + pf + extras_valu      # int: not before that gap
    gap, cnt_in_gap = 0, 0
    for ins in order:
        if isinstance(ins, int):
            if gap < ins:
                gap, cnt_in_gap = ins, 0
            continue
        if cnt_in_gap >= PER:
            gap, cnt_in_gap = gap + 1, 0
        while gap in (15, 31, 47):
            gap += 1
        if gap > 46:
            raise SystemExit("extras do not fit")
        after[gap].append(ins)
        cnt_in_gap += 1
    gap, cnt_in_gap = 49, 0
    for ins in late:
        if cnt_in_gap >= PER:
            gap, cnt_in_gap = gap + 1, 0
        after[gap].append(ins)
        cnt_in_gap += 1
    # refills
    for cb in range(4):
        if not (ABL & 1):
            after[16 * cb + 15].append(f"global_load_dwordx4 {v(AN + 4 * cb, 4)}, {v(V['xaddr'], 2)}, off offset:{64 * cb}")
    # ---- emit, computing vmcnt for blocks 1..3 from the issue order
    issued = 0

    def emit_gap(i):
        nonlocal issued
        for ins in after[i]:
            lines.append(ins)
            if is_vmem(ins):
                issued += 1

    emit_gap(-1)
    m_idx = 0
    for cb in range(4):
        if cb > 0:
            lines += timed_wait(f"s_waitcnt vmcnt({(3 - cb) + issued})", 100)
        for s2 in range(4):
            for ct in range(4):
                a = wcur + 16 * cb + 4 * s2 + ct
                b = AN + 4 * cb + s2
                d = acc + 4 * ct
                c = CIN + 4 * ct if (cb == 0 and s2 == 0) else d
                lines.append(f"v_mfma_f32_16x16x4_f32 {v(d, 4)}, {v(a)}, {v(b)}, {v(c, 4)}")
                emit_gap(m_idx)
                m_idx += 1
    lines.append(f"s_cmp_lg_u32 {s(grem)}, 0")
    return lines


def flush(P):
    """The sums of a step's last group leave for LDS (12 wait states behind the last MFMA)."""
    acc = ACCS[P]
    return ["s_nop 10"] + [f"ds_write_b128 {v(V['addr_prev'])}, {v(acc + 4 * ct, 4)}" + (f" offset:{16 * ct}" if ct else "")
                           for ct in range(4)]


def step_code(cur, nxt, name, other):
    """One step with the weights in buffer `cur` (the next step's go to `nxt`); falls through / jumps to step `other`."""
    L = []
    t, t2 = S['t'], S['t2']
    L.append(f"L_step{name}_%=:")
    L += stamp('top')
    # ---- sums of the first group (behind every earlier write in program order): their latency covers the bookkeeping
    L += [f"ds_read_b128 {v(CIN + 4 * ct, 4)}, {v(V['addr_cur'])} offset:{16 * ct}" for ct in range(4)]
    # ---- the next step's channel block and list; what the first group needs on entry (is it the last; the list entry of the
    # group behind it) goes out first: one LDS round trip that the rest of the bookkeeping covers
    L += [f"s_add_u32 {s(t)}, {s(S['sb'])}, 1",
          f"s_cmp_eq_u32 {s(t)}, {s(S['NSB'])}",
          f"s_cselect_b32 {s(S['last_sb'])}, 1, 0",
          f"s_cselect_b32 {s(S['nsb'])}, 0, {s(t)}",
          f"s_cselect_b32 {s(S['lb_ns'])}, {s(S['lb_nxt'])}, {s(S['lb_cur'])}",
          f"s_mov_b32 {s(S['grem'])}, {s(S['ng_cur'])}",
          f"s_add_u32 {s(t)}, {s(S['lb_cur'])}, 128",
          f"s_add_u32 {s(S['entry'])}, {s(S['lb_cur'])}, 256",
          f"s_cmp_eq_u32 {s(S['grem'])}, 1",
          f"s_cselect_b64 {s(S['lastm'], 2)}, -1, 0",
          f"s_cselect_b32 {s(t)}, {s(S['lb_ns'])}, {s(t)}",
          f"v_add_u32 {v(V['t0'])}, {s(t)}, {v(V['lst8'])}",
          f"ds_read_b64 {v(V['e'], 2)}, {v(V['t0'])}",
          # weights of the next step: same offset, next block — or the next offset (the last step re-reads its own)
          f"s_mul_i32 {s(t2)}, {s(S['nsb'])}, {s(S['cout4'])}",
          f"s_lshl_b32 {s(t2)}, {s(t2)}, 6",
          f"s_add_u32 {s(t)}, {s(S['k'])}, 1",
          f"s_cmp_lt_u32 {s(t)}, {s(S['K3'])}",
          f"s_cselect_b32 {s(t)}, {s(S['wk4'])}, 0",
          f"s_cmp_eq_u32 {s(S['last_sb'])}, 1",
          f"s_cselect_b32 {s(t)}, {s(t)}, {s(t2)}",
          f"s_add_u32 {s(S['Wl'])}, {s(S['Wk'])}, {s(t)}",
          f"s_addc_u32 {s(S['Wl'] + 1)}, {s(S['Wk'] + 1)}, 0",
          # rows: X + 256 sb (+ 16 q per lane), this step's and the next one's
          f"s_lshl_b32 {s(S['xoffc'])}, {s(S['sb'])}, 8",
          f"s_lshl_b32 {s(S['xoffn'])}, {s(S['nsb'])}, 8",
          f"s_add_u32 {s(S['t64'])}, {s(S['X'])}, {s(S['xoffc'])}",
          f"s_addc_u32 {s(S['t64'] + 1)}, {s(S['X'] + 1)}, 0",
          f"v_mov_b32 {v(V['xbc'] + 1)}, {s(S['t64'] + 1)}",
          f"v_add_co_u32_e32 {v(V['xbc'])}, vcc, {s(S['t64'])}, {v(V['xq'])}",
          f"v_addc_co_u32_e32 {v(V['xbc'] + 1)}, vcc, 0, {v(V['xbc'] + 1)}, vcc",
          f"s_add_u32 {s(S['t64'])}, {s(S['X'])}, {s(S['xoffn'])}",
          f"s_addc_u32 {s(S['t64'] + 1)}, {s(S['X'] + 1)}, 0",
          f"v_mov_b32 {v(V['xbn'] + 1)}, {s(S['t64'] + 1)}",
          f"v_add_co_u32_e32 {v(V['xbn'])}, vcc, {s(S['t64'])}, {v(V['xq'])}",
          f"v_addc_co_u32_e32 {v(V['xbn'] + 1)}, vcc, 0, {v(V['xbn'] + 1)}, vcc",
          f"s_cmp_eq_u32 {s(S['ng_cur'])}, 0",
          f"s_cbranch_scc1 L_empty{name}_%="]
    L += stamp('top')
    L += [f"s_cmp_eq_u32 {s(S['sb'])}, 0",
          f"s_cbranch_scc1 L_f2{name}_%="]
    L += group(cur, nxt, 1, 0, False)
    L += stamp('F', 'nF') + ([f"s_cmp_lg_u32 {s(S['grem'])}, 0"] if STAMPS else [])
    L += [f"s_cbranch_scc1 L_plain1{name}_%=", f"s_branch L_flush0{name}_%="]
    L.append(f"L_f2{name}_%=:")
    L += group(cur, nxt, 2, 0, False)
    L += stamp('F', 'nF') + ([f"s_cmp_lg_u32 {s(S['grem'])}, 0"] if STAMPS else [])
    L += [f"s_cbranch_scc0 L_flush0{name}_%="]
    L.append(f"L_plain1{name}_%=:")
    L += group(cur, nxt, 0, 1, True)
    L += stamp('P', 'nP') + ([f"s_cmp_lg_u32 {s(S['grem'])}, 0"] if STAMPS else [])
    L += [f"s_cbranch_scc0 L_flush1{name}_%="]
    L += group(cur, nxt, 0, 0, True)
    L += stamp('P', 'nP') + ([f"s_cmp_lg_u32 {s(S['grem'])}, 0"] if STAMPS else [])
    L += [f"s_cbranch_scc1 L_plain1{name}_%="]
    L.append(f"L_flush0{name}_%=:")
    L += flush(0)
    L += [f"s_branch L_end{name}_%="]
    L.append(f"L_flush1{name}_%=:")
    L += flush(1)
    L += [f"s_branch L_end{name}_%="]
    # ---- a step without a pair in this tile: its bookkeeping, exposed
    L.append(f"L_empty{name}_%=:")
    L += [f"s_cmp_eq_u32 {s(S['sb'])}, 0", f"s_cbranch_scc0 L_empty_w{name}_%="]
    L += [f"s_add_u32 {s(S['t'])}, {s(S['k'])}, 2",
          f"s_cmp_lt_u32 {s(S['t'])}, {s(S['K3'])}",
          f"s_cselect_b64 {s(S['kv'], 2)}, -1, 0",
          "s_waitcnt vmcnt(0)"]
    L += compaction(V['nv0'], V['nv1'], S['lb_nn'], S['ng_nn'], S['kv'])
    L += nv_loads(V['nv0'], V['nv1'])
    L.append(f"L_empty_w{name}_%=:")
    L += w_loads(nxt)
    # rows and sums' address of the next step's first group (valid entries even when it has none)
    L += [f"v_add_u32 {v(V['t0'])}, {s(S['lb_ns'])}, {v(V['lst8'])}",
          f"ds_read_b64 {v(V['e'], 2)}, {v(V['t0'])}",
          f"s_waitcnt lgkmcnt(0)",
          f"v_mad_u32_u24 {v(V['addr_cur'])}, {v(V['e'] + 1)}, {s(S['c272'])}, {v(V['ysq'])}",
          f"v_mad_u64_u32 {v(V['xaddr'], 2)}, vcc, {v(V['e'])}, {s(S['ldx4'])}, {v(V['xbn'], 2)}"]
    L += [f"global_load_dwordx4 {v(AN + 4 * cb, 4)}, {v(V['xaddr'], 2)}, off offset:{64 * cb}" for cb in range(4)]
    # ---- end of step: next channel block or next offset
    L.append(f"L_end{name}_%=:")
    L += [f"s_cmp_eq_u32 {s(S['last_sb'])}, 1",
          f"s_cbranch_scc0 L_samek{name}_%=",
          f"s_mov_b32 {s(S['ng_cur'])}, {s(S['ng_nxt'])}",
          f"s_mov_b32 {s(S['ng_nxt'])}, {s(S['ng_nn'])}",
          f"s_mov_b32 {s(S['t'])}, {s(S['lb_cur'])}",
          f"s_mov_b32 {s(S['lb_cur'])}, {s(S['lb_nxt'])}",
          f"s_mov_b32 {s(S['lb_nxt'])}, {s(S['lb_nn'])}",
          f"s_mov_b32 {s(S['lb_nn'])}, {s(S['t'])}",
          f"s_add_u32 {s(S['k'])}, {s(S['k'])}, 1",
          f"s_add_u32 {s(S['Wk'])}, {s(S['Wk'])}, {s(S['wk4'])}",
          f"s_addc_u32 {s(S['Wk'] + 1)}, {s(S['Wk'] + 1)}, 0",
          f"s_mov_b32 {s(S['sb'])}, -1",
          f"L_samek{name}_%=:",
          f"s_add_u32 {s(S['sb'])}, {s(S['sb'])}, 1",
          f"s_sub_u32 {s(S['steps'])}, {s(S['steps'])}, 1",
          f"s_cmp_eq_u32 {s(S['steps'])}, 0",
          f"s_cbranch_scc1 L_done_%="]
    if other is not None:
        L.append(f"s_branch L_step{other}_%=")
    return L


def program():
    L = []
    # ---- inputs -> fixed registers
    L += [f"s_mov_b64 {s(S['X'], 2)}, %[x]",
          f"s_mov_b64 {s(S['Wk'], 2)}, %[w]",
          f"s_mov_b64 {s(S['N'], 2)}, %[nb]",
          f"s_mov_b64 {s(S['Nstep'], 2)}, %[nstep]",
          f"s_mov_b32 {s(S['ldx4'])}, %[ldx4]",
          f"s_mov_b32 {s(S['cout4'])}, %[cout4]",
          f"s_mov_b32 {s(S['wk4'])}, %[wk4]",
          f"s_mov_b32 {s(S['K3'])}, %[k3]",
          f"s_mov_b32 {s(S['NSB'])}, %[nsb]",
          f"s_mov_b64 {s(S['valid0'], 2)}, %[valid0]",
          f"s_mov_b64 {s(S['valid1'], 2)}, %[valid1]",
          f"s_mov_b32 {s(S['lds'])}, %[lds]",
          f"s_lshl_b32 {s(S['c13'])}, {s(S['cout4'])}, 4",          # 16 weight rows
          f"s_mul_i32 {s(S['steps'])}, {s(S['K3'])}, {s(S['NSB'])}",
          f"s_movk_i32 {s(S['c272'])}, 272",
          f"v_mov_b32 {v(V['lane'])}, %[lane]",
          f"v_mov_b32 {v(V['woff'])}, %[woff]",
          f"v_add_u32 {v(V['woff1'])}, {s(S['cout4'])}, {v(V['woff'])}",
          f"v_add_u32 {v(V['woff2'])}, {s(S['cout4'])}, {v(V['woff1'])}",
          f"v_add_u32 {v(V['woff3'])}, {s(S['cout4'])}, {v(V['woff2'])}",
          f"v_mov_b32 {v(V['rowc0'])}, %[rowc0]",
          f"v_mov_b32 {v(V['rowc1'])}, %[rowc1]",
          # lane constants: m = lane & 15, q = lane >> 4
          f"v_and_b32 {v(V['t0'])}, 15, {v(V['lane'])}",
          f"v_lshlrev_b32 {v(V['lst8'])}, 3, {v(V['t0'])}",
          f"v_lshrrev_b32 {v(V['t1'])}, 4, {v(V['lane'])}",
          f"v_lshlrev_b32 {v(V['xq'])}, 4, {v(V['t1'])}",
          f"v_lshlrev_b32 {v(V['ysq'])}, 6, {v(V['t1'])}",
          f"v_add_u32 {v(V['ysq'])}, {s(S['lds'])}, {v(V['ysq'])}",
          f"v_lshlrev_b32 {v(V['dump'])}, 3, {v(V['lane'])}",
          f"s_add_u32 {s(S['t'])}, {s(S['lds'])}, {LDS_DUMP}",
          f"v_add_u32 {v(V['dump'])}, {s(S['t'])}, {v(V['dump'])}",
          f"v_mov_b32 {v(V['pad'])}, 0",
          f"v_mov_b32 {v(V['pad'] + 1)}, {R_ROWS}",
          f"v_mov_b32 {v(V['rid0'])}, {v(V['lane'])}",
          f"v_add_u32 {v(V['rid1'])}, 64, {v(V['lane'])}",
          f"v_mov_b32 {v(V['i0'] + 1)}, {v(V['rid0'])}",
          f"v_mov_b32 {v(V['i1'] + 1)}, {v(V['rid1'])}",
          f"v_mov_b32 {v(V['i2'] + 1)}, {v(V['rid0'])}",
          f"v_mov_b32 {v(V['i3'] + 1)}, {v(V['rid1'])}",
          f"s_add_u32 {s(S['lb_cur'])}, {s(S['lds'])}, {LDS_LISTS}",
          f"s_add_u32 {s(S['lb_nxt'])}, {s(S['lb_cur'])}, {LP * 8}",
          f"s_add_u32 {s(S['lb_nn'])}, {s(S['lb_nxt'])}, {LP * 8}",
          f"s_mov_b32 {s(S['k'])}, 0",
          f"s_mov_b32 {s(S['sb'])}, 0",
          f"s_mov_b32 {s(S['koff'])}, 0",
          f"s_mov_b32 {s(S['ng_nn'])}, 0"]
    if STAMPS:
        L += [f"s_mov_b64 {s(ST[x], 2)}, 0" for x in ('top', 'F', 'P')] + [f"s_mov_b32 {s(ST['nF'])}, 0", f"s_mov_b32 {s(ST['nP'])}, 0", "s_mov_b32 s100, 0", "s_mov_b32 s101, 0",
              f"s_memtime {s(ST['prev'], 2)}", "s_waitcnt lgkmcnt(0)"]
    # ---- neighbour rows of offsets 0, 1, 2; weights of step 0
    L += nv_loads(V['i0'], V['i1'])
    L += nv_loads(V['i2'], V['i3'])
    L += nv_loads(V['nv0'], V['nv1'])
    L += [f"s_mov_b64 {s(S['Wl'], 2)}, {s(S['Wk'], 2)}"]
    L += w_loads(WA)
    L += ["s_waitcnt vmcnt(18)"]          # offsets 0 and 1 (four loads) landed: 2 + 16 loads behind them
    L += compaction(V['i0'], V['i1'], S['lb_cur'], S['ng_cur'])
    L += compaction(V['i2'], V['i3'], S['lb_nxt'], S['ng_nxt'])
    # rows and sums' address of the very first group
    L += [f"v_add_u32 {v(V['t0'])}, {s(S['lb_cur'])}, {v(V['lst8'])}",
          f"ds_read_b64 {v(V['e'], 2)}, {v(V['t0'])}",
          f"v_mov_b32 {v(V['xbn'] + 1)}, {s(S['X'] + 1)}",
          f"v_add_co_u32_e32 {v(V['xbn'])}, vcc, {s(S['X'])}, {v(V['xq'])}",
          f"v_addc_co_u32_e32 {v(V['xbn'] + 1)}, vcc, 0, {v(V['xbn'] + 1)}, vcc",
          f"s_waitcnt lgkmcnt(0)",
          f"v_mad_u32_u24 {v(V['addr_cur'])}, {v(V['e'] + 1)}, {s(S['c272'])}, {v(V['ysq'])}",
          f"v_mad_u64_u32 {v(V['xaddr'], 2)}, vcc, {v(V['e'])}, {s(S['ldx4'])}, {v(V['xbn'], 2)}"]
    L += [f"global_load_dwordx4 {v(AN + 4 * cb, 4)}, {v(V['xaddr'], 2)}, off offset:{64 * cb}" for cb in range(4)]
    # ---- steps
    L += step_code(WA, WB, "A", "B")
    L += step_code(WB, WA, "B", "A")
    L += ["L_done_%=:", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    if STAMPS:
        L += stamp('top')
        L += [f"v_mov_b32 %[o{i}], {s(ST['top'] + i)}" for i in range(8)] + [f"v_mov_b32 %[o8], s100", f"v_mov_b32 %[o9], s101"]
    return L


def main():
    global ABL
    out = sys.argv[1] if len(sys.argv) > 1 else "cmp_asm.inc"
    variants = [int(x) for x in os.environ.get("CMA_VARIANTS", "").split(",") if x]
    L = program()
    clob = ["memory", "vcc", "scc"] + [f"v{i}" for i in range(0, NV_LAST + 1)] + [f"s{i}" for i in range(S_FIRST, (101 if STAMPS else S_LAST) + 1)]
    with open(out, "w") as f:
        f.write("// GENERATED by gen_cmp_asm.py - do not edit\n")
        f.write(f"#define CMA_LDS_BYTES {LDS_BYTES}\n#define CMA_LDS_LISTS {LDS_LISTS}\n#define CMA_LDS_DUMP {LDS_DUMP}\n"
                f"#define CMA_LDS_SROW {LDS_SROW}\n#define CMA_LP {LP}\n")
        f.write("#define CMA_ASM_TEXT \\\n")
        for ins in L:
            f.write(f'    "{ins}\\n\\t" \\\n')
        f.write('    ""\n')
        f.write("#define CMA_ASM_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
        for a in variants:       # experiment builds only (CMA_VARIANTS=1,2,...)
            ABL = a
            f.write(f"#define CMA_ASM_TEXT_V{a} \\\n")
            for ins in program():
                f.write(f'    "{ins}\\n\\t" \\\n')
            f.write('    ""\n')
        ABL = 0
        f.write("#define CMA_VARIANT_LIST " + " ".join(f"CMA_V({a})" for a in variants) + "\n")
    n_mfma = sum(1 for i in L if i.startswith("v_mfma"))
    print(f"{out}: {len(L)} instructions, {n_mfma} MFMAs", file=sys.stderr)


if __name__ == "__main__":
    main()
