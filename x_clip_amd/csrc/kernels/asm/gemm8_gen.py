#!/usr/bin/env python
"""Generator of the hand-scheduled gfx950 body of gemm8.h: the whole persistent tile loop of the bf16 ring GEMM as ONE asm unit.

    python x_clip_amd/csrc/kernels/asm/gemm8_gen.py          # rewrites gemm8_body.inc next to this file

Why a generator and not HIP: what gemm4.h's g5_run loses on the K = 512 products is the tile boundary (DESIGN_APPENDIX.md 6b: the epilogue of a tile is
not overlapped with matrix work, 15 % of such a launch), and the form that hides it -- a finished tile's packed output held in 64 registers and
stored two instructions per K step UNDER the next tile's MFMAs -- needs 128 accumulator + 48 fragment + 64 held registers live at once.  hipcc
spills that (profiles/r04_gemm_stage_i_iii_compile_evidence.txt); with every register named by hand it fits 246 of the 256.

What the body does (same LDS images, DMA pieces, barriers and counted waits as g5_run, whose C++ computes every per-lane address this
body uses -- gemm8.h passes them in as operands):
  * 8 waves of 128 x 64 on a 256 x 256 tile, K step 64 = 4 k-blocks of 8 MFMA 32x32x16; A in a ring of three 32 KiB LDS stages, B in two;
  * per K step: [kk0] fragment reads of kk1, 8 MFMAs with the four A pieces of step s + 2 behind the pairs; [kk1], [kk2] reads + MFMAs (+ the
    store slots of the previous tile); [kk3] vmcnt(4 + stores of this step), s_barrier, stage rotation, fragment reads of the NEXT stage's kk0,
    8 MFMAs with the four B pieces of step s + 2;
  * tile boundary: none.  Step 0 of a tile converts the previous tile's accumulators to bf16 in front of the C = 0 MFMA that overwrites each
    block (8 v_cvt_pk per MFMA), steps 0 .. 7 each swap + store one 32 x 32 block pair (row-per-lane 16-byte stores, T21), all other steps are
    the plain loop; the last tile of a work-group is drained after the loop.  The first tile's "previous tile" is a descriptor of size 0: its
    stores are dropped by the hardware.

Operands of the asm statement (gemm8.h must pass exactly these, in this order):
    %0 va0  %1 va1  %2 vb0  %3 vb1     per-lane byte offsets of the DMA pieces (g4_voff)
    %4 vaf  %5 vbf  %6 vbf1            per-lane fragment read addresses, stage-relative (k-major B: fragments I = 0 / 1; else %6 unused)
    %7 vc                              per-lane byte offset of the row-per-lane output stores
    %8 A  %9 B  %10 C                  64-bit bases
    %11 lda_b %12 ldb_b %13 ldc_b      leading dimensions in bytes
    %14 nt  %15 my_tiles               K steps per tile (>= 8), tiles this work-group walks (>= 1)
    %16 tm0 %17 nb0 %18 band0          first tile: row tile, column tile inside its band, band
    %19 tiles_m %20 band_n %21 dm %22 dn    tile-order constants (stride / 8 = dm * band_n + dn)
    %23 woff                           wave * 4096
    %24 ldsbase                        LDS address of the dynamic segment
    %25 vwr  %26 vrd                   per-lane addresses of the whole-line exchange inside the wave's 4 KiB slice (gemm4.h pack_lines_i)
(%7 vc is the row-per-lane store offset in the "rpl" variants and the whole-line store offset in the "lines" variants.)
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# ---- register map -------------------------------------------------------------------------------------------------------------------------
def ACC(b): return 16 * b                      # block b = 2 i + j: v[16 b .. 16 b + 15]
def FA(s, i): return 128 + 24 * s + 4 * i      # fragment set s: a0..a3
def FB(s, j): return 144 + 24 * s + 4 * j      #                 b0, b1
def HELD(b): return 176 + 8 * b                # packed bf16 of block b (8 registers)
TA, TB, TW, VACUR, VBCUR, VB1CUR = 240, 241, 242, 243, 244, 245
def NV(bk): return 246 if bk else 247          # v0 .. v245 (v246) are named here; the operands (10 / 9 of them) live above
def TXR(bk): return TB if bk else 246          # xor temporary of the exchange (k-major B leaves TB free; the other layout has one operand less)

RA, RB, RC = 40, 44, 48
A_CUR, A_NEXT, A_FREE, B_CUR, B_NEXT = 52, 53, 54, 55, 56
A_LEFT, B_LEFT, TILES_LEFT, GCNT = 57, 58, 59, 60
CUR_TM, CUR_TN, NXT_TM, NXT_NB, NXT_BAND, HAVE_PREV = 61, 62, 63, 64, 65, 66
SAOFF, SBOFF, SI1, SI2, SI3 = 67, 68, 69, 70, 71
T0, T1, T2, T3, T4, T5 = 72, 73, 74, 75, 76, 77
WOFF, NTM1 = 80, 81
PA, PB, PC = 82, 84, 86
ANT, BNT = 88, 90
BSTEP, CSIZE = 92, 94
SK1, SK2, SK3 = 95, 96, 97                     # 8 / 16 / 24 output rows in bytes
PREV2 = 98                                     # (spread stores) the step before a tile's first left two line stores behind its B pieces
S_LO, S_HI = 40, 99

OP = dict(va0="%0", va1="%1", vb0="%2", vb1="%3", vaf="%4", vbf="%5", vbf1="%6", vc="%7", A="%8", B="%9", C="%10", lda="%11", ldb="%12",
          ldc="%13", nt="%14", mytiles="%15", tm0="%16", nb0="%17", band0="%18", tilesm="%19", bandn="%20", dm="%21", dn="%22", woff="%23",
          ldsbase="%24", vwr="%25", vrd="%26")


def v(r, n=1): return f"v{r}" if n == 1 else f"v[{r}:{r + n - 1}]"
def s(r, n=1): return f"s{r}" if n == 1 else f"s[{r}:{r + n - 1}]"


class Body:
    def __init__(self, bk, ntst, sps=2, stores=True, cvt=True, mode="lines", early=False, spread=False, st_kk=3):
        self.bk, self.ntst, self.sps, self.do_stores, self.do_cvt, self.mode, self.early = bk, ntst, sps, stores, cvt, mode, early
        self.spread, self.st_kk = spread, st_kk
        self.L = []
        self.nlabel = 0

    def e(self, line): self.L.append(line)
    def c(self, text): self.L.append("; " + text)
    def label(self, stem):
        self.nlabel += 1
        return f"L8_{stem}_{self.nlabel}_%="

    # ---- pieces ---------------------------------------------------------------------------------------------------------------------------
    def read_frags(self, fs, va, vb, vb1, kk):
        """fragments of one k-block into set fs.  NT: `va` / `vb` already carry the k-block (xor); k-major B: kk is an immediate on vb / vb1"""
        if not self.bk:
            self.e(f"ds_read_b128 {v(FB(fs, 0), 4)}, {v(vb)}")
            self.e(f"ds_read_b128 {v(FB(fs, 1), 4)}, {v(vb)} offset:4096")
        else:
            off = kk * 2048
            self.e(f"ds_read_b64_tr_b16 {v(FB(fs, 0), 2)}, {v(vb)} offset:{off}")
            self.e(f"ds_read_b64_tr_b16 {v(FB(fs, 0) + 2, 2)}, {v(vb)} offset:{off + 512}")
            self.e(f"ds_read_b64_tr_b16 {v(FB(fs, 1), 2)}, {v(vb1)} offset:{off}")
            self.e(f"ds_read_b64_tr_b16 {v(FB(fs, 1) + 2, 2)}, {v(vb1)} offset:{off + 512}")
        for i in range(4):
            self.e(f"ds_read_b128 {v(FA(fs, i), 4)}, {v(va)}" + (f" offset:{4096 * i}" if i else ""))

    def mfma(self, b, fs, first):
        i, j = b >> 1, b & 1
        c = "0" if first else v(ACC(b), 16)
        self.e(f"v_mfma_f32_32x32x16_bf16 {v(ACC(b), 16)}, {v(FB(fs, j), 4)}, {v(FA(fs, i), 4)}, {c}")

    def dma_piece(self, which, q, base):
        """piece q of operand `which` into the LDS slice at SGPR `base` (+ q KiB)"""
        r = RA if which == "a" else RB
        vo = OP[("va" if which == "a" else "vb") + str(q & 1)]
        so = s(SAOFF if which == "a" else SBOFF) if q >> 1 else "0"
        if q == 0:
            self.e(f"s_mov_b32 m0, {s(base)}")
        else:
            self.e(f"s_add_u32 m0, {s(base)}, {1024 * q}")
        self.e("s_nop 0")
        self.e(f"buffer_load_dwordx4 {vo}, {s(r, 4)}, {so} offen lds")

    def adv(self, which):
        """DMA iterator of operand `which`: one K step on, or (branch-free) to the first step of the next tile"""
        r, left, nt_ = (RA, A_LEFT, ANT) if which == "a" else (RB, B_LEFT, BNT)
        step = "128" if which == "a" else s(BSTEP)
        self.e(f"s_add_u32 {s(T0)}, {s(r)}, {step}")
        self.e(f"s_addc_u32 {s(T1)}, {s(r + 1)}, 0")
        self.e(f"s_sub_u32 {s(T2)}, {s(left)}, 1")
        self.e(f"s_cmp_eq_u32 {s(left)}, 0")
        self.e(f"s_cselect_b32 {s(r)}, {s(nt_)}, {s(T0)}")
        self.e(f"s_cselect_b32 {s(r + 1)}, {s(nt_ + 1)}, {s(T1)}")
        self.e(f"s_cselect_b32 {s(left)}, {s(NTM1)}, {s(T2)}")

    def cvt_block(self, b):
        if not self.do_cvt:
            return
        for q in range(4):
            for h in range(2):
                self.e(f"v_cvt_pk_bf16_f32 {v(HELD(b) + 2 * q + h)}, {v(ACC(b) + 4 * q + 2 * h)}, {v(ACC(b) + 4 * q + 2 * h + 1)}")

    def swap_block(self, b):
        if not self.do_cvt:
            return
        self.e("s_nop 1")
        for x, y in ((0, 2), (1, 3), (4, 6), (5, 7)):
            self.e(f"v_permlane32_swap_b32 {v(HELD(b) + x)}, {v(HELD(b) + y)}")

    def store(self, b, o):
        if not self.do_stores:
            self.e("s_nop 0")
            return
        i, j = b >> 1, b & 1
        so = ["0", s(SI1), s(SI2), s(SI3)][i]
        imm = 64 * j + 32 * o
        self.e(f"buffer_store_dwordx4 {v(HELD(b) + 4 * o, 4)}, {OP['vc']}, {s(RC, 4)}, {so} offen" + (f" offset:{imm}" if imm else "") +
               (" nt" if self.ntst else ""))

    # ---- whole-line form: a 32-row group through the wave's 4 KiB slice (gemm4.h pack_lines_i / store_lines) ----------------------------------
    def group_writes(self, g, part):
        """half `part` (0 / 1 = block j) of group g: four 8-byte writes; TW = vwr + slice base must be set"""
        j = part
        for q in range(4):
            c = 4 * j + q
            src = v(HELD(2 * g + j) + 2 * q, 2)
            if c == 0:
                self.e(f"ds_write_b64 {v(TW)}, {src}")
            else:
                self.e(f"v_xor_b32 {v(TXR(self.bk))}, {16 * c}, {v(TW)}")
                self.e(f"ds_write_b64 {v(TXR(self.bk))}, {src}")

    def group_reads(self, g):
        self.e(f"v_add_u32 {v(TW)}, {s(T4)}, {OP['vrd']}")
        for k in range(4):
            self.e(f"ds_read_b128 {v(HELD(2 * g) + 4 * k, 4)}, {v(TW)}" + (f" offset:{1024 * k}" if k else ""))

    def group_stores(self, g, ks=(0, 1, 2, 3)):
        for k in ks:
            if not self.do_stores:
                self.e("s_nop 0")
                continue
            if g == 0:
                so = "0" if k == 0 else s(SK1 + k - 1)
            elif k == 0:
                so = s(SI1 + g - 1)
            else:
                self.e(f"s_add_u32 {s(T5)}, {s(SI1 + g - 1)}, {s(SK1 + k - 1)}")
                so = s(T5)
            self.e(f"buffer_store_dwordx4 {v(HELD(2 * g) + 4 * k, 4)}, {OP['vc']}, {s(RC, 4)}, {so} offen" + (" nt" if self.ntst else ""))

    def step_lines(self, first, group, prev_stores, tag):
        """lines mode.  first: C = 0 MFMAs behind the conversions; group: the 32-row group of the previous tile exchanged in kk1 (its A pieces
        wait for that: kk2) and stored behind the B pieces of kk3; prev_stores: line stores the step before left behind its B pieces"""
        self.c(f"---- K step ({tag}) ----")
        for kk in range(4):
            cur, nxt = kk & 1, (kk & 1) ^ 1
            self.c(f"k-block {kk}")
            if kk < 3:
                x = 32 * (kk + 1)
                self.e(f"v_xor_b32 {v(TA)}, {x}, {v(VACUR)}")
                if not self.bk:
                    self.e(f"v_xor_b32 {v(TB)}, {x}, {v(VBCUR)}")
                    self.read_frags(nxt, TA, TB, None, kk + 1)
                else:
                    self.read_frags(nxt, TA, VBCUR, VB1CUR, kk + 1)
                if kk == 0:
                    self.e(f"s_add_u32 {s(T4)}, {s(A_FREE)}, {s(WOFF)}")
                if kk == 1 and group is not None and self.do_cvt:
                    self.e(f"v_add_u32 {v(TW)}, {s(T4)}, {OP['vwr']}")
            else:
                self.e(f"s_waitcnt vmcnt({4 + (prev_stores if self.do_stores else 0)})")
                self.e("s_barrier")
                self.e(f"s_mov_b32 {s(T0)}, {s(A_CUR)}")
                self.e(f"s_mov_b32 {s(A_CUR)}, {s(A_NEXT)}")
                self.e(f"s_mov_b32 {s(A_NEXT)}, {s(A_FREE)}")
                self.e(f"s_mov_b32 {s(A_FREE)}, {s(T0)}")
                self.e(f"s_mov_b32 {s(T0)}, {s(B_CUR)}")
                self.e(f"s_mov_b32 {s(B_CUR)}, {s(B_NEXT)}")
                self.e(f"s_mov_b32 {s(B_NEXT)}, {s(T0)}")
                self.e(f"v_add_u32 {v(VACUR)}, {s(A_CUR)}, {OP['vaf']}")
                self.e(f"v_add_u32 {v(VBCUR)}, {s(B_CUR)}, {OP['vbf']}")
                if self.bk:
                    self.e(f"v_add_u32 {v(VB1CUR)}, {s(B_CUR)}, {OP['vbf1']}")
                self.read_frags(nxt, VACUR, VBCUR, VB1CUR, 0)
                self.e(f"s_add_u32 {s(T3)}, {s(B_NEXT)}, {s(WOFF)}")
            a_kk = 0 if (group is None or self.early) else 2
            for i in range(4):
                for j in range(2):
                    b = 2 * i + j
                    if first and kk == 0:
                        self.cvt_block(b)
                    self.mfma(b, cur, first and kk == 0)
                if kk == a_kk:
                    self.dma_piece("a", i, T4)
                if kk == 3:
                    self.dma_piece("b", i, T3)
                if kk == 1 and group is not None and self.do_cvt:
                    if i == 0: self.group_writes(group, 0)
                    if i == 1: self.group_writes(group, 1)
                    if i == 2: self.group_reads(group)
            if kk == a_kk:
                self.adv("a")
            if kk == 3:
                self.adv("b")
                if group is not None:
                    self.group_stores(group)
            self.e("s_waitcnt lgkmcnt(0)")

    # ---- the scheduled K step: fragment reads, DMA pieces and the exchange spread over the gaps between the MFMAs, counted lgkmcnt ----------
    def sb_reset(self, names):
        """LDS scoreboard at a step's entry: `names` were the last LDS operations issued, in this order (nothing is known to have completed)"""
        self.sb_ids = {n: k for k, n in enumerate(names)}
        self.sb_issued = len(names)
        self.sb_done = 0

    def sb_issue(self, name):
        self.sb_ids[name] = self.sb_issued
        self.sb_issued += 1

    def sb_need(self, names, extra=""):
        """the named LDS results must have landed (the LDS serves a wave's operations in order; lgkmcnt is a 4-bit counter)"""
        ks = [self.sb_ids[n] for n in names if n in self.sb_ids]
        if not ks or max(ks) < self.sb_done:
            if extra:
                self.e(f"s_waitcnt {extra}")
            return
        k = max(ks)
        n = self.sb_issued - 1 - k
        self.sb_done = k + 1
        if n <= 15:
            self.e(f"s_waitcnt {extra + ' ' if extra else ''}lgkmcnt({n})")
        elif extra:
            self.e(f"s_waitcnt {extra}")

    def frag_ops(self, fs, va, vb, vb1, kk):
        """[(name, instruction)] of the fragment reads of one k-block into set fs, in the order the MFMAs need them"""
        A = [(f"a{i}", f"ds_read_b128 {v(FA(fs, i), 4)}, {v(va)}" + (f" offset:{4096 * i}" if i else "")) for i in range(4)]
        if not self.bk:
            B = [[("b0", f"ds_read_b128 {v(FB(fs, 0), 4)}, {v(vb)}")], [("b1", f"ds_read_b128 {v(FB(fs, 1), 4)}, {v(vb)} offset:4096")]]
        else:
            off = kk * 2048
            B = [[("b0lo", f"ds_read_b64_tr_b16 {v(FB(fs, 0), 2)}, {v(vb)} offset:{off}"),
                  ("b0", f"ds_read_b64_tr_b16 {v(FB(fs, 0) + 2, 2)}, {v(vb)} offset:{off + 512}")],
                 [("b1lo", f"ds_read_b64_tr_b16 {v(FB(fs, 1), 2)}, {v(vb1)} offset:{off}"),
                  ("b1", f"ds_read_b64_tr_b16 {v(FB(fs, 1) + 2, 2)}, {v(vb1)} offset:{off + 512}")]]
        ops = B[0] + [A[0]] + B[1] + A[1:]
        return [(f"f{fs}.{n}", ins) for n, ins in ops]

    def step_sched(self, first, group, prev_stores, tag, stores=None, flag_wait=False):
        """as step_lines, with every k-block's work placed in the gaps behind its MFMAs and waits counted per operand.  Entry state (every path
        into a step provides it): the fragments of set 0 for k-block 0 were the last LDS operations issued, in frag_ops order."""
        self.c(f"---- K step ({tag}) ----")
        self.sb_reset([n for n, _ in self.frag_ops(0, 0, 0, 0, 0)])
        exch = group is not None and self.do_cvt
        for kk in range(4):
            cur, nxt = kk & 1, (kk & 1) ^ 1
            self.c(f"k-block {kk}")
            gaps = [[] for _ in range(9)]                       # gaps[0]: in front of the first MFMA; gaps[m + 1]: behind MFMA m
            if kk < 3:
                x = 32 * (kk + 1)
                gaps[0].append(("i", f"v_xor_b32 {v(TA)}, {x}, {v(VACUR)}"))
                if not self.bk:
                    gaps[0].append(("i", f"v_xor_b32 {v(TB)}, {x}, {v(VBCUR)}"))
                    reads = self.frag_ops(nxt, TA, TB, None, kk + 1)
                else:
                    reads = self.frag_ops(nxt, TA, VBCUR, VB1CUR, kk + 1)
                if kk == 0:
                    gaps[0].append(("i", f"s_add_u32 {s(T4)}, {s(A_FREE)}, {s(WOFF)}"))
            else:
                # all of this wave's reads of the current stages are through, DMA(s + 1) has landed: for every wave behind the barrier
                if self.st_kk == 2:
                    self.sb_need(list(self.sb_ids), f"vmcnt({4 + (len(stores) if (stores and self.do_stores) else 0)})")
                elif flag_wait and self.do_stores:
                    # (a tile's first step: what the step before it left behind its B pieces depends on the run time K)
                    la, lb = self.label("w4"), self.label("wdone")
                    self.e(f"s_cmp_eq_u32 {s(PREV2)}, 0")
                    self.e(f"s_cbranch_scc1 {la}")
                    self.sb_need(list(self.sb_ids), "vmcnt(6)")
                    self.e(f"s_branch {lb}")
                    self.e(f"{la}:")
                    self.e("s_waitcnt vmcnt(4) lgkmcnt(0)")
                    self.e(f"{lb}:")
                else:
                    self.sb_need(list(self.sb_ids), f"vmcnt({4 + (prev_stores if self.do_stores else 0)})")
                self.e("s_barrier")
                for ins in (f"s_mov_b32 {s(T0)}, {s(A_CUR)}", f"s_mov_b32 {s(A_CUR)}, {s(A_NEXT)}", f"s_mov_b32 {s(A_NEXT)}, {s(A_FREE)}",
                            f"s_mov_b32 {s(A_FREE)}, {s(T0)}", f"s_mov_b32 {s(T0)}, {s(B_CUR)}", f"s_mov_b32 {s(B_CUR)}, {s(B_NEXT)}",
                            f"s_mov_b32 {s(B_NEXT)}, {s(T0)}", f"v_add_u32 {v(VACUR)}, {s(A_CUR)}, {OP['vaf']}",
                            f"v_add_u32 {v(VBCUR)}, {s(B_CUR)}, {OP['vbf']}"):
                    gaps[0].append(("i", ins))
                if self.bk:
                    gaps[0].append(("i", f"v_add_u32 {v(VB1CUR)}, {s(B_CUR)}, {OP['vbf1']}"))
                gaps[0].append(("i", f"s_add_u32 {s(T3)}, {s(B_NEXT)}, {s(WOFF)}"))
                reads = self.frag_ops(nxt, VACUR, VBCUR, VB1CUR, 0)
            # fragment reads: all issued by the gap behind MFMA 5 (two per gap first when there are eight)
            spots = [1, 2, 3, 4, 5, 6] if len(reads) == 6 else [1, 1, 2, 2, 3, 4, 5, 6]
            for (n, ins), g in zip(reads, spots):
                gaps[g].append(("r", n, ins))
            a_kk = 0 if (group is None or self.early) else 2
            if kk == a_kk:
                for i in range(4):
                    gaps[2 * i + 2].append(("a", i))
            if kk == 3:
                for i in range(4):
                    gaps[2 * i + 2].append(("b", i))
            if kk == 1 and exch:
                gaps[0].append(("i", f"v_add_u32 {v(TW)}, {s(T4)}, {OP['vwr']}"))
                for c in range(8):
                    gaps[1 + c // 2].append(("w", c))
                gaps[5].append(("xr", None))
                for k in range(4):
                    gaps[5 + k].append(("x", k))

            if kk == 2 and stores and self.st_kk == 2:
                for n, (g, k) in enumerate(stores):
                    gaps[5 + 3 * n if len(stores) == 2 else 2 * n + 2].append(("s", g, k))

            def flush(gap):
                for it in gap:
                    if it[0] == "i":
                        self.e(it[1])
                    elif it[0] == "r":
                        self.e(it[2])
                        self.sb_issue(it[1])
                    elif it[0] == "a":
                        if exch and not self.early and it[1] == 0:
                            self.sb_need([f"x{k}" for k in range(4)])        # the slice is about to be overwritten
                        self.dma_piece("a", it[1], T4)
                    elif it[0] == "b":
                        self.dma_piece("b", it[1], T3)
                    elif it[0] == "w":
                        c = it[1]
                        src = v(HELD(2 * group + (c >> 2)) + 2 * (c & 3), 2)
                        if c == 0:
                            self.e(f"ds_write_b64 {v(TW)}, {src}")
                        else:
                            self.e(f"v_xor_b32 {v(TXR(self.bk))}, {16 * c}, {v(TW)}")
                            self.e(f"ds_write_b64 {v(TXR(self.bk))}, {src}")
                        self.sb_issue(f"w{c}")
                    elif it[0] == "s":
                        if exch:
                            self.sb_need([f"x{k}" for k in range(4)])
                        self.group_stores(it[1], (it[2],))
                    elif it[0] == "xr":
                        self.e(f"v_add_u32 {v(TW)}, {s(T4)}, {OP['vrd']}")
                    elif it[0] == "x":
                        k = it[1]
                        self.e(f"ds_read_b128 {v(HELD(2 * group) + 4 * k, 4)}, {v(TW)}" + (f" offset:{1024 * k}" if k else ""))
                        self.sb_issue(f"x{k}")

            flush(gaps[0])
            for m in range(8):
                i, j = m >> 1, m & 1
                if first and kk == 0:
                    self.cvt_block(m)
                self.sb_need([f"f{cur}.a{i}", f"f{cur}.b{j}"])
                self.mfma(m, cur, first and kk == 0)
                flush(gaps[m + 1])
            if kk == a_kk:
                self.adv("a")
            if kk == 3:
                self.adv("b")
                if stores is None and group is not None:
                    stores = [(group, k) for k in range(4)]
                if stores and self.st_kk == 3:
                    if exch:
                        self.sb_need([f"x{k}" for k in range(4)])
                    for (g, k) in stores:
                        self.group_stores(g, (k,))

    def next_bases(self):
        """first-step addresses of the `next` tile's operands -> ANT, BNT"""
        self.e(f"s_lshl_b32 {s(T0)}, {s(NXT_TM)}, 8")
        self.e(f"s_mul_i32 {s(T1)}, {s(T0)}, {OP['lda']}")
        self.e(f"s_mul_hi_u32 {s(T2)}, {s(T0)}, {OP['lda']}")
        self.e(f"s_add_u32 {s(ANT)}, {s(PA)}, {s(T1)}")
        self.e(f"s_addc_u32 {s(ANT + 1)}, {s(PA + 1)}, {s(T2)}")
        self.e(f"s_mul_i32 {s(T0)}, {s(NXT_BAND)}, {OP['bandn']}")
        self.e(f"s_add_u32 {s(T0)}, {s(T0)}, {s(NXT_NB)}")
        self.e(f"s_lshl_b32 {s(T0)}, {s(T0)}, 8")
        if not self.bk:
            self.e(f"s_mul_i32 {s(T1)}, {s(T0)}, {OP['ldb']}")
            self.e(f"s_mul_hi_u32 {s(T2)}, {s(T0)}, {OP['ldb']}")
        else:
            self.e(f"s_lshl_b32 {s(T1)}, {s(T0)}, 1")
            self.e(f"s_mov_b32 {s(T2)}, 0")
        self.e(f"s_add_u32 {s(BNT)}, {s(PB)}, {s(T1)}")
        self.e(f"s_addc_u32 {s(BNT + 1)}, {s(PB + 1)}, {s(T2)}")

    def c_rsrc_from_cur(self):
        """RC <- the output tile of `cur` (size 0 while there is no finished tile)"""
        self.e(f"s_lshl_b32 {s(T0)}, {s(CUR_TM)}, 8")
        self.e(f"s_mul_i32 {s(T1)}, {s(T0)}, {OP['ldc']}")
        self.e(f"s_mul_hi_u32 {s(T2)}, {s(T0)}, {OP['ldc']}")
        self.e(f"s_lshl_b32 {s(T3)}, {s(CUR_TN)}, 9")
        self.e(f"s_add_u32 {s(T1)}, {s(T1)}, {s(T3)}")
        self.e(f"s_addc_u32 {s(T2)}, {s(T2)}, 0")
        self.e(f"s_add_u32 {s(RC)}, {s(PC)}, {s(T1)}")
        self.e(f"s_addc_u32 {s(RC + 1)}, {s(PC + 1)}, {s(T2)}")
        self.e(f"s_cmp_eq_u32 {s(HAVE_PREV)}, 0")
        self.e(f"s_cselect_b32 {s(RC + 2)}, 0, {s(CSIZE)}")

    def tile_start(self):
        self.c("tile start: the finished tile's output descriptor, cur <- next, next <- the tile after (if any), its operand addresses")
        self.c_rsrc_from_cur()
        self.e(f"s_mov_b32 {s(CUR_TM)}, {s(NXT_TM)}")
        self.e(f"s_mul_i32 {s(T0)}, {s(NXT_BAND)}, {OP['bandn']}")
        self.e(f"s_add_u32 {s(CUR_TN)}, {s(T0)}, {s(NXT_NB)}")
        self.e(f"s_cmp_eq_u32 {OP['nt']}, 8")
        self.e(f"s_cselect_b32 {s(PREV2)}, {s(HAVE_PREV)}, 0")
        self.e(f"s_mov_b32 {s(HAVE_PREV)}, 1")
        noadv, a1, a2 = self.label("noadv"), self.label("adv1"), self.label("adv2")
        self.e(f"s_cmp_gt_u32 {s(TILES_LEFT)}, 1")
        self.e(f"s_cbranch_scc0 {noadv}")
        self.e(f"s_add_u32 {s(NXT_NB)}, {s(NXT_NB)}, {OP['dn']}")
        self.e(f"s_cmp_ge_u32 {s(NXT_NB)}, {OP['bandn']}")
        self.e(f"s_cbranch_scc0 {a1}")
        self.e(f"s_sub_u32 {s(NXT_NB)}, {s(NXT_NB)}, {OP['bandn']}")
        self.e(f"s_add_u32 {s(NXT_TM)}, {s(NXT_TM)}, 1")
        self.e(f"{a1}:")
        self.e(f"s_add_u32 {s(NXT_TM)}, {s(NXT_TM)}, {OP['dm']}")
        self.e(f"{a2}:")
        self.e(f"s_cmp_ge_u32 {s(NXT_TM)}, {OP['tilesm']}")
        self.e(f"s_cbranch_scc0 {noadv}")
        self.e(f"s_sub_u32 {s(NXT_TM)}, {s(NXT_TM)}, {OP['tilesm']}")
        self.e(f"s_add_u32 {s(NXT_BAND)}, {s(NXT_BAND)}, 1")
        self.e(f"s_branch {a2}")
        self.e(f"{noadv}:")
        self.next_bases()

    # ---- one K step ------------------------------------------------------------------------------------------------------------------------
    def step(self, first, swap_blocks, stores, tag):
        """first: C = 0 MFMAs behind the conversion of each block of the previous tile; swap_blocks: blocks whose halves are exchanged in kk1;
        stores: [(block, half)] issued in kk1 / kk2"""
        self.c(f"---- K step ({tag}) ----")
        st = list(stores)
        per_kk = {1: st[:(len(st) + 1) // 2], 2: st[(len(st) + 1) // 2:]}
        for kk in range(4):
            cur, nxt = kk & 1, (kk & 1) ^ 1
            self.c(f"k-block {kk}")
            if kk < 3:
                x = 32 * (kk + 1)
                self.e(f"v_xor_b32 {v(TA)}, {x}, {v(VACUR)}")
                if not self.bk:
                    self.e(f"v_xor_b32 {v(TB)}, {x}, {v(VBCUR)}")
                    self.read_frags(nxt, TA, TB, None, kk + 1)
                else:
                    self.read_frags(nxt, TA, VBCUR, VB1CUR, kk + 1)
                if kk == 0:
                    self.e(f"s_add_u32 {s(T4)}, {s(A_FREE)}, {s(WOFF)}")
            else:
                self.e(f"s_waitcnt vmcnt({4 + (len(st) if self.do_stores else 0)})")
                self.e("s_barrier")
                self.e(f"s_mov_b32 {s(T0)}, {s(A_CUR)}")
                self.e(f"s_mov_b32 {s(A_CUR)}, {s(A_NEXT)}")
                self.e(f"s_mov_b32 {s(A_NEXT)}, {s(A_FREE)}")
                self.e(f"s_mov_b32 {s(A_FREE)}, {s(T0)}")
                self.e(f"s_mov_b32 {s(T0)}, {s(B_CUR)}")
                self.e(f"s_mov_b32 {s(B_CUR)}, {s(B_NEXT)}")
                self.e(f"s_mov_b32 {s(B_NEXT)}, {s(T0)}")
                self.e(f"v_add_u32 {v(VACUR)}, {s(A_CUR)}, {OP['vaf']}")
                self.e(f"v_add_u32 {v(VBCUR)}, {s(B_CUR)}, {OP['vbf']}")
                if self.bk:
                    self.e(f"v_add_u32 {v(VB1CUR)}, {s(B_CUR)}, {OP['vbf1']}")
                self.read_frags(nxt, VACUR, VBCUR, VB1CUR, 0)
                self.e(f"s_add_u32 {s(T4)}, {s(B_NEXT)}, {s(WOFF)}")
            slot = list(per_kk.get(kk, []))
            for i in range(4):
                for j in range(2):
                    b = 2 * i + j
                    if first and kk == 0:
                        self.cvt_block(b)
                    self.mfma(b, cur, first and kk == 0)
                if kk == 0:
                    self.dma_piece("a", i, T4)
                if kk == 3:
                    self.dma_piece("b", i, T4)
                if kk == 1 and i == 0:
                    for b in swap_blocks:
                        self.swap_block(b)
                if kk in (1, 2) and i >= 1 and slot:
                    # (spread over the MFMA pairs behind the first)
                    n = (len(slot) + (3 - i)) // (4 - i)
                    for _ in range(n):
                        self.store(*slot.pop(0))
            if kk == 0:
                self.adv("a")
            if kk == 3:
                self.adv("b")
            self.e("s_waitcnt lgkmcnt(0)")

    # ---- the whole body -------------------------------------------------------------------------------------------------------------------
    def build(self):
        e = self.e
        self.c("gemm8 body: generated by gemm8_gen.py -- do not edit")
        e(f"s_mov_b64 {s(PA, 2)}, {OP['A']}")
        e(f"s_mov_b64 {s(PB, 2)}, {OP['B']}")
        e(f"s_mov_b64 {s(PC, 2)}, {OP['C']}")
        e(f"s_lshl_b32 {s(SAOFF)}, {OP['lda']}, 4")
        e(f"s_lshl_b32 {s(SBOFF)}, {OP['ldb']}, 4")
        e(f"s_lshl_b32 {s(SI1)}, {OP['ldc']}, 5")
        e(f"s_lshl_b32 {s(SI2)}, {OP['ldc']}, 6")
        e(f"s_add_u32 {s(SI3)}, {s(SI2)}, {s(SI1)}")
        e(f"s_lshl_b32 {s(SK1)}, {OP['ldc']}, 3")
        e(f"s_lshl_b32 {s(SK2)}, {OP['ldc']}, 4")
        e(f"s_add_u32 {s(SK3)}, {s(SK2)}, {s(SK1)}")
        e(f"s_sub_u32 {s(NTM1)}, {OP['nt']}, 1")
        e(f"s_mov_b32 {s(NXT_TM)}, {OP['tm0']}")
        e(f"s_mov_b32 {s(NXT_NB)}, {OP['nb0']}")
        e(f"s_mov_b32 {s(NXT_BAND)}, {OP['band0']}")
        e(f"s_mov_b32 {s(CUR_TM)}, 0")
        e(f"s_mov_b32 {s(CUR_TN)}, 0")
        e(f"s_mov_b32 {s(TILES_LEFT)}, {OP['mytiles']}")
        e(f"s_mov_b32 {s(HAVE_PREV)}, 0")
        e(f"s_mov_b32 {s(WOFF)}, {OP['woff']}")
        e(f"s_mul_i32 {s(T0)}, {OP['lda']}, 255")
        e(f"s_add_u32 {s(RA + 2)}, {s(T0)}, 128")
        e(f"s_mov_b32 {s(RA + 3)}, 0x00020000")
        if not self.bk:
            e(f"s_mul_i32 {s(T0)}, {OP['ldb']}, 255")
            e(f"s_add_u32 {s(RB + 2)}, {s(T0)}, 128")
            e(f"s_mov_b32 {s(BSTEP)}, 128")
        else:
            e(f"s_mul_i32 {s(T0)}, {OP['ldb']}, 63")
            e(f"s_add_u32 {s(RB + 2)}, {s(T0)}, 512")
            e(f"s_lshl_b32 {s(BSTEP)}, {OP['ldb']}, 6")
        e(f"s_mov_b32 {s(RB + 3)}, 0x00020000")
        e(f"s_mul_i32 {s(T0)}, {OP['ldc']}, 255")
        e(f"s_add_u32 {s(CSIZE)}, {s(T0)}, 512")
        e(f"s_mov_b32 {s(RC + 2)}, 0")
        e(f"s_mov_b32 {s(RC + 3)}, 0x00020000")
        self.next_bases()
        e(f"s_mov_b32 {s(RA)}, {s(ANT)}")
        e(f"s_mov_b32 {s(RA + 1)}, {s(ANT + 1)}")
        e(f"s_mov_b32 {s(RB)}, {s(BNT)}")
        e(f"s_mov_b32 {s(RB + 1)}, {s(BNT + 1)}")
        e(f"s_mov_b32 {s(A_LEFT)}, {s(NTM1)}")
        e(f"s_mov_b32 {s(B_LEFT)}, {s(NTM1)}")
        e(f"s_mov_b32 {s(A_CUR)}, {OP['ldsbase']}")
        e(f"s_add_u32 {s(A_NEXT)}, {s(A_CUR)}, 0x8000")
        e(f"s_add_u32 {s(A_FREE)}, {s(A_CUR)}, 0x10000")
        e(f"s_add_u32 {s(B_CUR)}, {s(A_CUR)}, 0x18000")
        e(f"s_add_u32 {s(B_NEXT)}, {s(A_CUR)}, 0x20000")
        self.c("prologue: A(0), B(0), A(1), B(1); the first two must have landed before step 0")
        e(f"s_add_u32 {s(T4)}, {s(A_CUR)}, {s(WOFF)}")
        for q in range(4): self.dma_piece("a", q, T4)
        e(f"s_add_u32 {s(T4)}, {s(B_CUR)}, {s(WOFF)}")
        for q in range(4): self.dma_piece("b", q, T4)
        self.adv("a")
        self.adv("b")
        e(f"s_add_u32 {s(T4)}, {s(A_NEXT)}, {s(WOFF)}")
        for q in range(4): self.dma_piece("a", q, T4)
        e(f"s_add_u32 {s(T4)}, {s(B_NEXT)}, {s(WOFF)}")
        for q in range(4): self.dma_piece("b", q, T4)
        self.adv("a")
        self.adv("b")
        e("s_waitcnt vmcnt(8)")
        e("s_barrier")
        e(f"v_add_u32 {v(VACUR)}, {s(A_CUR)}, {OP['vaf']}")
        e(f"v_add_u32 {v(VBCUR)}, {s(B_CUR)}, {OP['vbf']}")
        if self.bk:
            e(f"v_add_u32 {v(VB1CUR)}, {s(B_CUR)}, {OP['vbf1']}")
        self.read_frags(0, VACUR, VBCUR, VB1CUR, 0)
        e("s_waitcnt lgkmcnt(0)")

        ltile, lgen, lgend = self.label("tile"), self.label("gen"), self.label("genend")
        e(f"{ltile}:")
        self.tile_start()
        if self.mode == "rpl":
            nspecial = 16 // self.sps
            for t in range(nspecial):
                blocks = [b for b in range(8) if (2 * b) // self.sps == t]
                stores = [(b, o) for b in blocks for o in (0, 1)]
                self.step(t == 0, blocks, stores, f"tile step {t}: previous tile's blocks {blocks}")
        elif self.spread:
            # two line stores behind the B pieces of each of the tile's first eight steps: group g's lines 0, 1 in step g (its exchange is in
            # that step's kk1), lines 2, 3 in step g + 4; a ninth step (if the tile has one) still counts the two stores of the eighth
            nspecial = 8
            for t in range(8):
                self.step_sched(t == 0, t if t < 4 else None, 2, f"tile step {t}", stores=[(t & 3, 2 * (t >> 2)), (t & 3, 2 * (t >> 2) + 1)],
                                flag_wait=(t == 0))
        else:
            nspecial = 5
            st = self.step_sched if self.mode == "sched" else self.step_lines
            for t in range(4):
                st(t == 0, t, 0 if t == 0 else 4, f"tile step {t}: previous tile's row group {t}")
            st(False, None, 4, "tile step 4: plain, behind a step with line stores")
        e(f"s_sub_u32 {s(GCNT)}, {OP['nt']}, {nspecial}")
        if self.mode == "sched" and self.spread:
            e(f"s_cmp_eq_u32 {s(GCNT)}, 0")
            e(f"s_cbranch_scc1 {lgend}")
            self.step_sched(False, None, 2, "tile step 8: plain, behind a step with two line stores")
            e(f"s_sub_u32 {s(GCNT)}, {s(GCNT)}, 1")
        e(f"{lgen}:")
        e(f"s_cmp_eq_u32 {s(GCNT)}, 0")
        e(f"s_cbranch_scc1 {lgend}")
        if self.mode == "rpl":
            self.step(False, [], [], "plain")
        elif self.mode == "sched":
            self.step_sched(False, None, 0, "plain")
        else:
            self.step_lines(False, None, 0, "plain")
        e(f"s_sub_u32 {s(GCNT)}, {s(GCNT)}, 1")
        e(f"s_branch {lgen}")
        e(f"{lgend}:")
        e(f"s_sub_u32 {s(TILES_LEFT)}, {s(TILES_LEFT)}, 1")
        e(f"s_cmp_lg_u32 {s(TILES_LEFT)}, 0")
        e(f"s_cbranch_scc1 {ltile}")
        self.c("drain: the last tile")
        self.c_rsrc_from_cur()
        if self.mode == "rpl":
            for b in range(8):
                self.cvt_block(b)
                self.swap_block(b)
                self.store(b, 0)
                self.store(b, 1)
        else:
            for b in range(8):
                self.cvt_block(b)
            e(f"s_add_u32 {s(T4)}, {s(A_FREE)}, {s(WOFF)}")
            for g in range(4):
                if self.do_cvt:
                    e(f"v_add_u32 {v(TW)}, {s(T4)}, {OP['vwr']}")
                    self.group_writes(g, 0)
                    self.group_writes(g, 1)
                    self.group_reads(g)
                    e("s_waitcnt lgkmcnt(0)")
                self.group_stores(g)
        e("s_waitcnt vmcnt(0)")
        return self.L


def clobbers(bk):
    # scc (every s_add / s_cmp of the body) is named so that the unit stays safe if code is ever placed behind it.  m0 (the
    # LDS-DMA base) cannot be named: it is a RESERVED register to the compiler ("inline asm clobber list contains reserved registers: m0"),
    # which never keeps a value in it across statements and re-materialises it in front of every instruction of its own that reads it
    return [f"v{i}" for i in range(NV(bk))] + [f"s{i}" for i in range(S_LO, S_HI + 1)] + ["vcc", "scc", "memory"]


def c_string(lines):
    out = []
    for l in lines:
        if l.startswith(";"):
            out.append(f"    /* {l[2:]} */")
        else:
            out.append('    "' + l + '\\n\\t"')
    return "\n".join(out)


# measurement variants (gemm8.h instantiates 0 / 1 in the product, all of them in the measurement build): id -> Body options
VARIANTS = {
    0: dict(ntst=False, mode="sched", spread=True),   # product: scheduled K step, whole-line stores through the LDS slice, two per K step
    1: dict(ntst=True, mode="sched", spread=True),    # product: the same, non-temporal (outputs the L2s cannot hold)
    2: dict(ntst=True, mode="rpl", sps=2),         # row-per-lane stores, two per K step (first version: -20 ... -45 %)
    3: dict(ntst=True, mode="lines"),              # whole-line stores, k-blocks as g5_run orders them (reads, then MFMAs, then lgkmcnt(0))
    4: dict(ntst=True, mode="sched", spread=True, st_kk=2),   # the two line stores of a step in its kk2 (between the A and the B pieces)
    5: dict(ntst=True, mode="sched", spread=True, stores=False),              # (garbage) conversions and exchange, no stores
    6: dict(ntst=True, mode="sched", spread=True, stores=False, cvt=False),   # (garbage) the bare loop
}


def main():
    parts = ["// gemm8_body.inc -- GENERATED by gemm8_gen.py (hand-scheduled gfx950 body of gemm8_kernel); do not edit.",
             "// G8_BODY_<layout>_<variant>: layout NT (B row-major [N, K]) / NN (B k-major [K, N]); variants: gemm8_gen.py VARIANTS.", ""]
    for bk in (False, True):
        for var, opt in VARIANTS.items():
            name = f"G8_BODY_{'NN' if bk else 'NT'}_{var}"
            body = Body(bk, **opt).build()
            n_inst = sum(1 for l in body if not l.startswith(";") and not l.endswith(":"))
            if var >= 2:
                parts.append("#ifdef XCLIP_MEASURE")
            parts.append(f"// {name}: {opt}, {n_inst} instructions")
            parts.append(f"#define {name} \\")
            lines = c_string(body).split("\n")
            parts.append(" \\\n".join(lines))
            if var >= 2:
                parts.append("#endif")
            parts.append("")
    parts.append("#define G8_CLOBBERS_NT " + ", ".join(f'"{c}"' for c in clobbers(False)))
    parts.append("#define G8_CLOBBERS_NN " + ", ".join(f'"{c}"' for c in clobbers(True)))
    parts.append("#define G8_RPL_VARIANTS(V) ((V) == 2)")
    parts.append(f"#define G8_VARIANTS {len(VARIANTS)}")
    parts.append("")
    path, text = os.path.join(HERE, "gemm8_body.inc"), "\n".join(parts)
    if os.path.exists(path) and open(path).read() == text:     # (untouched when nothing changed: the libraries are rebuilt by file time)
        print("unchanged", path)
        return
    with open(path, "w") as f:
        f.write(text)
    print("wrote", path)


if __name__ == "__main__":
    main()
