#!/usr/bin/env python3
"""Generator of the hand-scheduled K-loop of gemm_a4_kernel (v2s_gemm_a4.inc is its output; build.sh runs it).

Why a generator: the kernel keeps ONE wave per SIMD (4 waves, 256 x 256 block tile, 128 x 128 wave tile = 256 fp32 accumulators in
AGPRs, v_mfma_f32_32x32x16_bf16), so nothing hides a stall: every ds_read / LDS-DMA / scalar instruction has to sit in the shadow of
an MFMA and every wait has to be a COUNTED s_waitcnt.  hipcc does not schedule that (its one-wave loops stop at
1.0-1.1 PF with 48 % issue stalls), so the whole K loop is one inline-asm statement with literal registers, and this script is the
scheduler: it places the filler instructions in the MFMA gaps and derives every s_waitcnt count from a scoreboard of the wave's
outstanding LDS / vector-memory operations instead of by hand.

Replaces the main loop of every nn.Linear forward / dgrad GEMM with M, N >= 256 (reference: model/modeling_t5.py:304-311,528-536,581;
model/vit.py:41,53,17,20).

Structure (per wave; "stage" = 32 k of the block tile = 16 KiB of A rows + 16 KiB of B rows in LDS, ring of 4 stages = 128 KiB;
"step" = 16 k = one v_mfma_32x32x16 per accumulator block = 16 MFMAs):
  step 2s   (even): MFMAs on fragment set F0 | ds_reads of (stage s, k-half 1) -> F1 | LDS-DMA of stage s+3 into the slot of stage s-1
  step 2s+1 (odd) : MFMAs on F1 | after the 2nd MFMA: s_waitcnt vmcnt(stage s+1 landed) lgkmcnt(0); s_barrier | ds_reads of
                    (stage s+1, k-half 0) -> F0
  The barrier in step 2s+1 publishes stage s+1 (RAW) and proves every wave is done reading stage s (WAR for the DMA of stage s+4,
  issued in step 2s+2).  The DMA stream never stops: past the last stage it re-requests the last stage, so every count is a constant.

LDS images (the DMA writes 1 KiB per wave-instruction at M0 + 16 * lane, so any swizzle is applied to the per-lane SOURCE address):
  K-contiguous operand ([rows][32 k], 64 B per row): 16-byte slot g of row r is stored at slot g ^ ((r >> 2) & 3); read with ds_read_b128
  by lane (h = lane >> 5, r = lane & 31): conflict-free for the b128 lane groups.
  [k][n] operand (dgrad's W; TB): [32 k][256 n], 512 B per k-row, 32-byte granule g of k-row k stored at granule g ^ 2 (k & 3); read with
  ds_read_b64_tr_b16 (two per fragment).
Accumulator orientation: mfma(srcA = W rows (n), srcB = X rows (m)): lane (h, m) of block (bi, bj) holds output row m and sixteen
columns; the W rows of a fragment are read in the permuted order n = 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3) (i = MFMA row index), which
makes those sixteen columns CONSECUTIVE: n = 32 bj + 16 h + r for accumulator register r.
"""
import sys

# ---- literal register map (one wave) -----------------------------------------------------------------------------------------
F = [0, 32]                  # fragment sets: BF[bj] = v[F + 4 bj .. +3], AF[bi] = v[F + 16 + 4 bi .. +3]
V_OA, V_OB = 64, 68          # per-lane DMA source offsets (4 + 4)
V_DA = 72                    # ds_read addresses of the A operand, 8 registers.  K-contiguous operand: [k-half 0, k-half 1, k-half 0 + 64 KiB,
V_DB = 80                    # k-half 1 + 64 KiB]; [k][rows] operand (tr reads): [block 0..3] and [block 0..3] + 64 KiB.  Same for B.
V_T = 88                     # temporaries v88..v103
S_PA, S_PB = 36, 38          # running source bases (64-bit)
S_LDSW = 40                  # LDS base + wave * 4096
S_IT = 41                    # loop counter
S_STA, S_STB = 42, 43        # bytes per stage
S_ADV = 44                   # stages the pointers may still advance
S_T = 46                     # temporaries s46..s55
NV_CLOBBER = 104             # v0..v103
STAGE = 32768
A_PART, B_PART = 0, 16384


# ---- slot plan of one stage (two steps of 16 MFMA gaps): fragment reads and LDS-DMA requests are SPREAD, never clustered -- the issue
# of an LDS-DMA request costs the wave 60-185 cycles when it follows other requests / LDS reads closely, ~25-60 when it stands alone
# (MI355X_MICROARCH.md, per-instruction constants), and the MFMA behind it only hides 32.  A stage's 8 requests therefore go out one
# per ~4 MFMAs over 30 gaps: requests 0-3 of stage s+4 in the odd step of stage s (after its barrier), requests 4-7 in the even step
# of stage s+1; the M0 write sits one gap ahead of its request (the MFMA between them is the wait state it needs).
import os
_PLAN = os.environ.get("A4_PLAN", "spread")
if _PLAN == "early":         # fragment reads as early as the step allows (complete long before the next wait), requests in the remaining gaps
    EVEN_READS = [0, 1, 2, 3, 4, 5, 6, 7]
    EVEN_REQ = [9, 11, 13, 15]
    ODD_READS = [2, 3, 4, 5, 6, 7, 8, 9]
    ODD_REQ = [11, 12, 14, 15]
else:
    EVEN_READS = [0, 1, 3, 5, 7, 9, 11, 13]
    EVEN_REQ = [2, 6, 10, 14]
    ODD_READS = [2, 3, 5, 6, 8, 9, 11, 12]
    ODD_REQ = [4, 7, 10, 14]


STORE_MOD = (" " + os.environ["A4_STORE_MOD"]) if os.environ.get("A4_STORE_MOD") else ""      # cache policy of the output stores (experiment: "sc1", "nt", "sc0 sc1")
ZSPREAD = int(os.environ.get("A4_ZSPREAD", "0"))      # (dact form, experiment) > 0: the z loads go out one per ZSPREAD gaps over TWO iterations instead of back to back (K >= 640)
BG_UNIT = int(os.environ.get("A4_BG_UNIT", "3"))     # background-stream budget of one MFMA gap (units; one full-rate VALU instruction = 1)


class Gen:
    def __init__(self, tb, ta=False, abl=""):
        self.tb, self.ta = tb, ta
        self.abl = abl           # profiling ablations of the main loop (results invalid): "nodma" = no LDS-DMA requests, "noread" = no fragment
                                 # reads, "none" = neither (MFMAs + barriers only), "nobar" = no s_barrier
        self.lines = []
        self.lgkm = []       # outstanding LDS operations, oldest first: tags
        self.vm = []         # outstanding vector-memory operations: tags
        self.nmfma = 0

    def e(self, s):
        if self.abl == "nobar" and s == "s_barrier":
            return
        if self.abl in ("nodma", "none") and s.startswith("s_add_u32 m0,"):
            return
        self.lines.append(s)

    # ---- scoreboard
    def lds_op(self, tag, text):
        if self.abl in ("noread", "none") and tag.startswith("F"):
            return
        if len(self.lgkm) >= 15:                # lgkmcnt is a 4-bit counter: retire the older half (long done) to make room
            keep = 7
            self.e(f"s_waitcnt lgkmcnt({keep})")
            self.lgkm = self.lgkm[len(self.lgkm) - keep:]
        self.lgkm.append(tag)
        self.e(text)

    def need(self, tags):
        """every LDS operation whose tag is in `tags` must have returned"""
        idx = max((i for i, t in enumerate(self.lgkm) if t in tags), default=-1)
        if idx >= 0:
            self.e(f"s_waitcnt lgkmcnt({len(self.lgkm) - idx - 1})")
            self.lgkm = self.lgkm[idx + 1:]

    def vm_op(self, tag, text):
        assert len(self.vm) < 63
        self.vm.append(tag)
        if self.abl in ("nodma", "none") and tag.startswith("stage"):
            return               # (the scoreboard still counts it: waits become vmcnt(N) on an emptier queue, i.e. free)
        self.e(text)

    def vm_need(self, tag, also_lgkm0=False):
        idx = max(i for i, t in enumerate(self.vm) if t == tag)
        n = len(self.vm) - idx - 1
        self.vm = self.vm[idx + 1:]
        if also_lgkm0:
            self.e("s_waitcnt lgkmcnt(0)" if self.abl == "nowait" else f"s_waitcnt vmcnt({n}) lgkmcnt(0)")
            self.lgkm = []
        else:
            self.e(f"s_waitcnt vmcnt({n})")

    # ---- pieces
    def frag_reads(self, slot, kh, fs):
        """ds_reads of (ring slot, k-half kh) into fragment set fs, in the order the MFMAs want them: BF0 AF0 BF1 BF2 BF3 AF1 AF2 AF3.
        Returns a list of (tag, text) groups (one group = the reads of one fragment)."""
        hi = slot >> 1
        base = (slot & 1) * STAGE

        def frag(which, blk):
            tr = self.ta if which == "A" else self.tb
            reg = F[fs] + (16 if which == "A" else 0) + 4 * blk
            vd, part = (V_DA, A_PART) if which == "A" else (V_DB, B_PART)
            tag = f"F{fs}{which}{blk}"
            if not tr:
                return [(tag, f"ds_read_b128 v[{reg}:{reg + 3}], v{vd + kh + 2 * hi} offset:{base + part + blk * 2048}")]
            off = base + part + kh * 8192
            return [(tag, f"ds_read_b64_tr_b16 v[{reg}:{reg + 1}], v{vd + blk + 4 * hi} offset:{off}"),
                    (tag, f"ds_read_b64_tr_b16 v[{reg + 2}:{reg + 3}], v{vd + blk + 4 * hi} offset:{off + 2048}")]

        return [frag(w, i) for w, i in (("B", 0), ("A", 0), ("B", 1), ("B", 2), ("B", 3), ("A", 1), ("A", 2), ("A", 3))]

    def dma_pairs(self, slot, tag):
        """the 8 LDS-DMA requests of one stage: (m0 setup, request)"""
        out = []
        for part, voff, sp in ((A_PART, V_OA, S_PA), (B_PART, V_OB, S_PB)):
            for i in range(4):
                imm = slot * STAGE + part + i * 1024
                out.append((f"s_add_u32 m0, s{S_LDSW}, 0x{imm:x}", (tag, f"global_load_lds_dwordx4 v{voff + i}, s[{sp}:{sp + 1}]")))
        return out

    def advance(self):
        t = S_T
        return [f"s_cmp_lg_u32 s{S_ADV}, 0",
                f"s_cselect_b32 s{t}, s{S_STA}, 0",
                f"s_cselect_b32 s{t + 1}, s{S_STB}, 0",
                f"s_cselect_b32 s{t + 2}, 1, 0",
                f"s_add_u32 s{S_PA}, s{S_PA}, s{t}",
                f"s_addc_u32 s{S_PA + 1}, s{S_PA + 1}, 0",
                f"s_add_u32 s{S_PB}, s{S_PB}, s{t + 1}",
                f"s_addc_u32 s{S_PB + 1}, s{S_PB + 1}, 0",
                f"s_sub_u32 s{S_ADV}, s{S_ADV}, s{t + 2}"]

    def mfma(self, fs, bi, bj, first=False):
        acc = 16 * (4 * bi + bj)
        bf, af = F[fs] + 4 * bj, F[fs] + 16 + 4 * bi
        self.need({f"F{fs}B{bj}", f"F{fs}A{bi}"})
        c = "0" if first else f"a[{acc}:{acc + 15}]"
        self.e(f"v_mfma_f32_32x32x16_bf16 a[{acc}:{acc + 15}], v[{bf}:{bf + 3}], v[{af}:{af + 3}], {c}")
        self.nmfma += 1

    def step(self, fs, gaps, first=False, pre=None):
        """16 MFMAs on fragment set fs; gaps[j] = list of callables run after MFMA j, pre[j] = before it; after the gap's own fillers
        up to self.bg_rate entries of the background stream self.bg (deferred write-out work) are emitted"""
        j = 0
        for bi in range(4):
            for bj in range(4):
                for f in (pre or {}).get(j, []):
                    f()
                self.mfma(fs, bi, bj, first)
                for f in gaps.get(j, []):
                    f()
                left = getattr(self, "bg_rate", 0) * BG_UNIT          # a write-out / load entry fills a gap's budget, a hash instruction costs 1-2 units
                while left > 0 and getattr(self, "bg", None):
                    ent = self.bg[0]
                    w, f = ent if isinstance(ent, tuple) else (BG_UNIT, ent)
                    if w > left:
                        break
                    self.bg.pop(0)
                    f()
                    left -= w
                j += 1

    # ---- address arithmetic shared by the one-tile and the persistent kernel.  v{V_T} = lane, s{S_T} = wave, s{S_T+1} = wm, s{S_T+2} = wn.
    def setup_common(self):
        e = self.e
        T, S = V_T, S_T
        e("s_nop 4")
        e(f"v_and_b32 v{T}, 63, %[tid]")                    # lane
        e(f"v_lshrrev_b32 v{T + 1}, 6, %[tid]")             # wave
        e("s_nop 1")
        e(f"v_readfirstlane_b32 s{S}, v{T + 1}")            # w
        e(f"s_lshr_b32 s{S + 1}, s{S}, 1")                  # wm
        e(f"s_and_b32 s{S + 2}, s{S}, 1")                   # wn
        e(f"s_lshl_b32 s{S + 3}, s{S}, 12")
        e(f"s_add_u32 s{S_LDSW}, %[lds], s{S + 3}")

    def setup_dma_rows(self, voff, ld, r0=None, rmax=None):
        """K-contiguous operand ([rows][32 k] image): chunk c = wave * 4 + i = 16 rows x 64 B; lane -> row (lane >> 2), stored slot lane & 3
        holds logical slot (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3).  Offsets relative to row r0 (clamped to rmax) or,
        when r0 is None, to the tile's first row (no clamp: whole tiles)."""
        e = self.e
        T, S = V_T, S_T
        e(f"v_lshrrev_b32 v{T + 2}, 2, v{T}")
        e(f"v_lshrrev_b32 v{T + 3}, 4, v{T}")
        e(f"v_and_b32 v{T + 3}, 3, v{T + 3}")
        e(f"v_and_b32 v{T + 4}, 3, v{T}")
        e(f"v_xor_b32 v{T + 4}, v{T + 4}, v{T + 3}")
        e(f"v_lshlrev_b32 v{T + 4}, 4, v{T + 4}")           # logical slot * 16 bytes
        e(f"s_lshl_b32 s{S + 4}, s{S}, 6")                  # w * 64 rows
        if r0 is not None:
            e(f"s_add_u32 s{S + 4}, {r0}, s{S + 4}")
        e(f"v_add_u32 v{T + 5}, s{S + 4}, v{T + 2}")
        for i in range(4):
            e(f"v_add_u32 v{T + 6}, {16 * i}, v{T + 5}")
            if rmax is not None:
                e(f"v_min_u32 v{T + 6}, {rmax}, v{T + 6}")
            e(f"v_mul_lo_u32 v{T + 6}, v{T + 6}, {ld}")
            e(f"v_add_u32 v{voff + i}, v{T + 6}, v{T + 4}")

    def setup_dma_tr(self, voff, ld, c0=None, cmax=None):
        """[k][rows] operand ([32 k][256 rows] image, 512 B per k-row): chunk c = wave * 4 + i holds k-rows 2c, 2c+1; lane -> k = 2c + (lane >> 5),
        stored 16-byte slot p = lane & 31 = granule p >> 1 (holds logical granule (p >> 1) ^ 2 (k & 3)), half p & 1; k & 3 = 2 (i & 1) + (lane >> 5)"""
        e = self.e
        T, S = V_T, S_T
        e(f"v_lshrrev_b32 v{T + 5}, 5, v{T}")               # lane >> 5
        e(f"v_and_b32 v{T + 6}, 31, v{T}")                  # p
        e(f"v_lshrrev_b32 v{T + 7}, 1, v{T + 6}")           # stored granule
        e(f"v_and_b32 v{T + 8}, 1, v{T + 6}")               # half
        e(f"s_lshl_b32 s{S + 5}, s{S}, 3")                  # w * 8
        e(f"v_add_u32 v{T + 9}, s{S + 5}, v{T + 5}")        # w * 8 + (lane >> 5)
        for i in range(4):
            e(f"v_add_u32 v{T + 10}, {2 * (i & 1)}, v{T + 5}")           # k & 3
            e(f"v_lshlrev_b32 v{T + 10}, 1, v{T + 10}")
            e(f"v_xor_b32 v{T + 10}, v{T + 7}, v{T + 10}")              # logical granule
            e(f"v_lshlrev_b32 v{T + 10}, 4, v{T + 10}")                 # * 16 columns
            e(f"v_lshl_add_u32 v{T + 10}, v{T + 8}, 3, v{T + 10}")      # + half * 8
            if c0 is not None:
                e(f"v_add_u32 v{T + 10}, {c0}, v{T + 10}")
                e(f"v_min_u32 v{T + 10}, {cmax}, v{T + 10}")
            e(f"v_lshlrev_b32 v{T + 10}, 1, v{T + 10}")                 # bytes
            e(f"v_add_u32 v{T + 11}, {2 * i}, v{T + 9}")                # k
            e(f"v_mul_lo_u32 v{T + 11}, v{T + 11}, {ld}")
            e(f"v_add_u32 v{voff + i}, v{T + 11}, v{T + 10}")

    def setup_ds_rows(self, vd, swh, perm):
        """ds_read_b128 addresses of a K-contiguous operand: lane (h, i): tile row wh * 128 + perm(i), slot (2 kh + h) ^ ((row >> 2) & 3).
        perm(i) = 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3) (the W rows: makes a lane's sixteen accumulator columns consecutive), (row >> 2) & 3 = i >> 3"""
        e = self.e
        T = V_T
        e(f"v_and_b32 v{T + 5}, 31, v{T}")                  # i
        e(f"v_lshrrev_b32 v{T + 6}, 5, v{T}")               # h
        if perm:
            e(f"v_lshrrev_b32 v{T + 7}, 2, v{T + 5}")
            e(f"v_and_b32 v{T + 7}, 1, v{T + 7}")
            e(f"v_lshlrev_b32 v{T + 7}, 4, v{T + 7}")
            e(f"v_lshrrev_b32 v{T + 8}, 3, v{T + 5}")       # i >> 3 (= f)
            e(f"v_lshl_add_u32 v{T + 7}, v{T + 8}, 2, v{T + 7}")
            e(f"v_and_b32 v{T + 9}, 3, v{T + 5}")
            e(f"v_add_u32 v{T + 7}, v{T + 7}, v{T + 9}")    # row = perm(i)
        else:
            e(f"v_mov_b32 v{T + 7}, v{T + 5}")              # row = i
            e(f"v_lshrrev_b32 v{T + 8}, 2, v{T + 5}")
            e(f"v_and_b32 v{T + 8}, 3, v{T + 8}")           # f
        e(f"v_xor_b32 v{T + 8}, v{T + 6}, v{T + 8}")        # h ^ f
        e(f"v_lshlrev_b32 v{T + 8}, 4, v{T + 8}")
        e(f"s_lshl_b32 s{S_T + 5}, s{swh}, 13")             # wh * 128 rows * 64 B
        e(f"v_lshl_add_u32 v{T + 7}, v{T + 7}, 6, s{S_T + 5}")
        e(f"v_add_u32 v{T + 7}, %[lds], v{T + 7}")
        e(f"v_add_u32 v{vd}, v{T + 7}, v{T + 8}")
        e(f"v_xor_b32 v{vd + 1}, 32, v{vd}")                # k-half 1: slot ^ 2 (the LDS base is a multiple of 64)
        e(f"v_add_u32 v{vd + 2}, 0x10000, v{vd}")
        e(f"v_add_u32 v{vd + 3}, 0x10000, v{vd + 1}")

    def setup_ds_tr(self, vd, swh, perm):
        """ds_read_b64_tr_b16 addresses of a [k][rows] operand: 16-lane group grp = lane >> 4 (k-half hh = grp >> 1, row half nh = grp & 1); lane j
        of a group supplies the address of 4 rows of k-row hh * 8 + (j >> 2) and receives row j of the 4 x 16 block the group's addresses
        describe.  Run cq = j & 3 of the block must be the MFMA indices 16 nh + 4 cq .. +3 of the 32-row fragment blk:
          natural order: rows 16 nh + 4 cq ..: granule (wh * 8 + 2 blk + nh) ^ 2 q = wh * 8 + nh + 2 (blk ^ q), byte 8 cq inside it   (q = j >> 2 = k & 3)
          permuted (W) : rows 16 (cq & 1) + 8 nh + 4 (cq >> 1) ..: granule wh * 8 + (cq & 1) + 2 (blk ^ q), byte 16 nh + 8 (cq >> 1)"""
        e = self.e
        T = V_T
        e(f"v_and_b32 v{T + 7}, 15, v{T}")                  # j
        e(f"v_lshrrev_b32 v{T + 8}, 4, v{T}")               # grp
        e(f"v_and_b32 v{T + 9}, 1, v{T + 8}")               # nh
        e(f"v_lshrrev_b32 v{T + 10}, 1, v{T + 8}")          # hh
        e(f"v_lshrrev_b32 v{T + 11}, 2, v{T + 7}")          # q
        e(f"v_lshl_add_u32 v{T + 12}, v{T + 10}, 3, v{T + 11}")         # k-row hh * 8 + q
        e(f"v_lshlrev_b32 v{T + 12}, 9, v{T + 12}")         # * 512 B
        e(f"s_lshl_b32 s{S_T + 5}, s{swh}, 3")              # wh * 8
        if perm:
            e(f"v_and_b32 v{T + 13}, 1, v{T + 7}")          # cq & 1
            e(f"v_add_u32 v{T + 13}, s{S_T + 5}, v{T + 13}")
            e(f"v_lshl_add_u32 v{T + 12}, v{T + 13}, 5, v{T + 12}")     # + granule base * 32
            e(f"v_lshl_add_u32 v{T + 12}, v{T + 9}, 4, v{T + 12}")      # + 16 nh
            e(f"v_lshrrev_b32 v{T + 13}, 1, v{T + 7}")
            e(f"v_and_b32 v{T + 13}, 1, v{T + 13}")
            e(f"v_lshl_add_u32 v{T + 12}, v{T + 13}, 3, v{T + 12}")     # + 8 (cq >> 1)
        else:
            e(f"v_add_u32 v{T + 13}, s{S_T + 5}, v{T + 9}")             # wh * 8 + nh
            e(f"v_lshl_add_u32 v{T + 12}, v{T + 13}, 5, v{T + 12}")
            e(f"v_and_b32 v{T + 13}, 3, v{T + 7}")                      # cq
            e(f"v_lshl_add_u32 v{T + 12}, v{T + 13}, 3, v{T + 12}")     # + 8 cq
        e(f"v_add_u32 v{T + 12}, %[lds], v{T + 12}")
        for blk in range(4):
            e(f"v_xor_b32 v{T + 13}, {blk}, v{T + 11}")                 # blk ^ q
            e(f"v_lshl_add_u32 v{vd + blk}, v{T + 13}, 6, v{T + 12}")
            e(f"v_add_u32 v{vd + 4 + blk}, 0x10000, v{vd + blk}")

    # ---- the statement
    def setup(self):
        e = self.e
        S = S_T
        self.setup_common()
        e(f"s_mov_b32 s{S_PA}, %[pa0]"); e(f"s_mov_b32 s{S_PA + 1}, %[pa1]")
        e(f"s_mov_b32 s{S_PB}, %[pb0]"); e(f"s_mov_b32 s{S_PB + 1}, %[pb1]")
        e(f"s_mov_b32 s{S_IT}, %[niter]")
        e(f"s_lshl_b32 s{S_ADV}, %[niter], 2")
        e(f"s_sub_u32 s{S_ADV}, s{S_ADV}, 1")               # nst - 1 pointer advances
        if not self.ta:
            e(f"s_mov_b32 s{S_STA}, 64")
            self.setup_dma_rows(V_OA, "%[lda]", "%[m0]", "%[mmax]")
            self.setup_ds_rows(V_DA, S + 1, perm=False)
        else:
            e(f"s_lshl_b32 s{S_STA}, %[lda], 5")            # 32 k-rows per stage
            self.setup_dma_tr(V_OA, "%[lda]", "%[m0]", "%[mmax]")
            self.setup_ds_tr(V_DA, S + 1, perm=False)
        if not self.tb:
            e(f"s_mov_b32 s{S_STB}, 64")
            self.setup_dma_rows(V_OB, "%[ldb]", "%[n0]", "%[nmax]")
            self.setup_ds_rows(V_DB, S + 2, perm=True)
        else:
            e(f"s_lshl_b32 s{S_STB}, %[ldb], 5")
            self.setup_dma_tr(V_OB, "%[ldb]", "%[n0]", "%[nmax]")
            self.setup_ds_tr(V_DB, S + 2, perm=True)
        e("s_nop 4")                                        # VALU-written VGPRs / SALU-written SGPRs -> vector memory

    def issue_stage_now(self, slot, tag):
        for m0set, (t, req) in self.dma_pairs(slot, tag):
            self.e(m0set)
            self.e("s_nop 0")
            self.vm_op(t, req)
        for s in self.advance():
            self.e(s)

    def fill_even(self, st, gaps):
        """even step of ring slot st: reads of (slot st, k-half 1) -> F1; requests 4-7 of the stage three ahead (slot (st + 3) & 3)"""
        reads = self.frag_reads(st, 1, 1)
        for g, grp in zip(EVEN_READS, reads):
            gaps.setdefault(g, []).extend((lambda tg=tg, tx=tx: self.lds_op(tg, tx)) for tg, tx in grp)
        pairs = self.dma_pairs((st + 3) & 3, f"stage{self.stage_ctr + 3}")[4:]
        for g, (m0set, (t, req)) in zip(EVEN_REQ, pairs):
            gaps.setdefault(g - 1, []).append(lambda m0set=m0set: self.e(m0set))
            gaps.setdefault(g, []).insert(0, (lambda t=t, req=req: self.vm_op(t, req)))

    def fill_odd(self, st, gaps, adv, extra_sync=None):
        """odd step of ring slot st: pointer bookkeeping, the stage's wait + barrier, reads of (slot st + 1, k-half 0) -> F0, requests 0-3 of
        the stage four ahead (slot st: every wave is past its last read of it once it is through the barrier)"""
        gaps.setdefault(0, []).extend((lambda x=x: self.e(x)) for x in adv)

        def sync():
            self.vm_need(f"stage{self.stage_ctr + 1}", also_lgkm0=True)
            self.e("s_barrier")
        gaps.setdefault(1, []).append(sync)
        if extra_sync:
            gaps[1].append(extra_sync)
        reads = self.frag_reads((st + 1) & 3, 0, 0)
        for g, grp in zip(ODD_READS, reads):
            gaps.setdefault(g, []).extend((lambda tg=tg, tx=tx: self.lds_op(tg, tx)) for tg, tx in grp)
        pairs = self.dma_pairs(st & 3, f"stage{self.stage_ctr + 4}")[:4]
        for g, (m0set, (t, req)) in zip(ODD_REQ, pairs):
            gaps.setdefault(g - 1, []).append(lambda m0set=m0set: self.e(m0set))
            gaps.setdefault(g, []).insert(0, (lambda t=t, req=req: self.vm_op(t, req)))

    def body(self, first_iter_c0):
        """4 stages = 8 steps; first_iter_c0: the MFMAs of stage 0 / k-half 0 start from C = 0 (peeled first iteration)"""
        for st in range(4):
            gaps = {}
            self.fill_even(st, gaps)
            self.step(0, gaps, first=(first_iter_c0 and st == 0))
            gaps = {}
            self.fill_odd(st, gaps, self.advance())
            self.step(1, gaps)
            self.stage_ctr += 1

    def prologue(self, advance):
        """stages 0, 1, 2 and the first half of stage 3 requested; stage 0 landed and published; the first fragments on their way"""
        e = self.e
        self.stage_ctr = 0
        for s in range(4):
            pairs = self.dma_pairs(s, f"stage{s}")
            for m0set, (t, req) in (pairs if s < 3 else pairs[:4]):
                e(m0set); e("s_nop 0"); self.vm_op(t, req)
            if s < 3:
                for x in advance():
                    e(x)
        self.vm_need("stage0")
        e("s_barrier")
        for grp in self.frag_reads(0, 0, 0):
            for tg, tx in grp:
                self.lds_op(tg, tx)

    def main(self):
        e = self.e
        self.setup()
        self.prologue(self.advance)
        # first iteration peeled (C = 0 in its first step), then the loop
        shape = lambda: (list(self.lgkm), len(self.vm))
        entry = shape()
        self.body(True)
        assert shape() == entry, (shape(), entry)
        e(f"s_sub_u32 s{S_IT}, s{S_IT}, 1")
        e(f"s_cmp_eq_u32 s{S_IT}, 0")
        e("s_cbranch_scc1 L_a4_done_%=")
        e("L_a4_loop_%=:")
        self.vm = [f"stage{self.stage_ctr + 1}"] * 8 + [f"stage{self.stage_ctr + 2}"] * 8 + [f"stage{self.stage_ctr + 3}"] * 4
        self.body(False)
        assert shape() == entry
        e(f"s_sub_u32 s{S_IT}, s{S_IT}, 1")
        e(f"s_cmp_lg_u32 s{S_IT}, 0")
        e("s_cbranch_scc1 L_a4_loop_%=")
        e("L_a4_done_%=:")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")
        return self.lines


# =====================================================================================================================================
# Persistent form with a DEFERRED write-out ("a4p"): one block per CU walks its tiles; the bf16 output of tile t leaves the chip during
# the main loop of tile t+1.  Plain bf16 epilogue (alpha = 1), M >= 256, N >= 512 (multiples of 8), K >= 384.
#   * step 0 of a tile: before the first MFMA of accumulator block b (which starts from C = 0) the block's sixteen values of the PREVIOUS
#     tile are read out of the AGPRs and rounded to bf16 pairs (v_cvt_pk_bf16_f32) into eight "held" VGPRs (128 in all);
#   * steps 1..8: a background stream of write-out work, one instruction per MFMA gap: per 32-row block row c of the wave's 128 x 128
#     sub-tile  W_c: 8 ds_write_b128 held -> a wave-private 8 KiB staging block (rows of 256 B, 16-byte chunks XOR-swizzled with
#     row & 7),  R_c: 8 ds_read_b128 back as whole rows (4 rows x 256 B per instruction, into the same held registers),  S_c: 8
#     buffer_store_dwordx4 (each lane 16 B, 16 lanes = one 256-byte row segment: full cache lines).  No barrier: the staging block is
#     private to the wave and LDS executes a wave's operations in order;
#   * the DMA stream does not stop at a tile edge: the last three stages' requests of a tile already fetch the next tile (scalar bases
#     switched by a uniform branch in the last loop iteration); past the block's last tile the same tile is requested again (harmless)
#     so that every counted wait keeps its constant;
#   * the stores ride the wave's in-order vmcnt queue between the DMA requests: the scoreboard counts them like any other operation.
#     The first tile of a block has nothing held: its stores go through a buffer descriptor with num_records = 0 (dropped).
#   * after the block's last tile: convert + write out without a main loop beside it (drain).
V_HELD = 104                 # v104..v231
V_STW = 232                  # staging write addresses: 4 variants q = 2 (bj & 1) + half
V_STR = 236                  # staging read addresses: row-group parity 0 / 1
V_STO = 238                  # store offset: (lane >> 4) * ldc_bytes + (lane & 15) * 16
V_ZO = 239                   # (dact form) load offset of the mask operand z in the accumulator layout: (lane & 31) * ldc_bytes + (lane >> 5) * 32
V_ZM = 240                   # (dact form) eight mask temporaries v240..v247
NV_CLOBBER_P = 250
S_K, S_NMY = 44, 45          # tile ordinal of this block, its tile count
S_SRD = 56                   # s56..s59 buffer descriptor of C
S_ST_PREV, S_ST_CUR, S_ST_NEXT = 60, 61, 62   # store offsets (bytes): running one of the held tile, base of the current / next tile
S_ST_WAVE = 63               # (wm * 128) * ldc_bytes + wn * 256
S_ST_STEP = 64               # 4 rows: 4 * ldc_bytes
S_X = 68                     # temporaries s68..s79 (tile coordinates)
S_ZSRD = 80                  # (dact form) s80..s83 buffer descriptor of z
S_SCALE = 84                 # (dact form) 1 / (1 - p) of the forward's dropout, fp32 bits
S_ZROW = 85                  # (dact form) s85..s88: z offsets of the wave's four block rows of the CURRENT tile
S_HK1, S_HK2 = 89, 90        # (reludrop form) the two multipliers of v2s_mix32
S_HXS = 91                   # 0x80008000: unsigned -> signed 16-bit draws
S_HPP = 92                   # both halves: ((p16 ^ 0x8000) - 1) & 0xffff
S_HC = 93                    # s93..s96: the four pair multipliers of v2s_keep8
NS_CLOBBER_P = 98
V_HL = 239                   # (reludrop form; = V_ZO's register) lane part of the chunk index + seed term
V_HB, V_HT = 240, 241        # hash temporaries
V_SC2 = 248                  # v248..v249: the dropout scale twice (v_pk_mul_f32 operand)
RING = 4 * STAGE


class GenP(Gen):
    def __init__(self, tb, epi=""):
        """epi: "" = plain; "dact" = the ReLU-mask epilogue of a dgrad (v2s_gemm dact = RELU with z = the forward's post-dropout activation):
        out = z > 0 ? acc * scale : 0.  The z tile of the CURRENT tile is prefetched during the tile's third iteration ("Z") straight into the
        128 held registers, in the accumulator layout (row per lane, 32 contiguous bytes per block), long before the conversion needs it;
        the conversion multiplies by the scale in fp32, rounds, and ANDs with a per-half mask derived from z (sat16(0 - z) >> 15: all ones
        iff the bf16 is > 0).  Requires ldz == ldc and K >= 512."""
        super().__init__(tb)
        self.epi = epi
        self.bg = []
        self.bg_rate = 1

    # ---- lane constants (tile-independent)
    def setup_p(self):
        e = self.e
        T, S = V_T, S_T
        self.setup_common()
        e(f"s_mov_b32 s{S_STA}, 64")
        self.setup_dma_rows(V_OA, "%[lda]")
        self.setup_ds_rows(V_DA, S + 1, perm=False)
        if not self.tb:
            e(f"s_mov_b32 s{S_STB}, 64")
            self.setup_dma_rows(V_OB, "%[ldb]")
            self.setup_ds_rows(V_DB, S + 2, perm=True)
        else:
            e(f"s_lshl_b32 s{S_STB}, %[ldb], 5")
            self.setup_dma_tr(V_OB, "%[ldb]")
            self.setup_ds_tr(V_DB, S + 2, perm=True)
        # ---- write-out constants.  Staging block of this wave: LDS + ring + wave * 8 KiB, 32 rows x 256 B.
        # lane (h, m) holds row m, 16-byte chunks 4 bj + 2 h + half of the row; chunk c of row m is stored at chunk c ^ (m & 7)
        e(f"s_lshl_b32 s{S + 5}, s{S}, 13")
        e(f"s_add_u32 s{S + 5}, s{S + 5}, 0x{RING:x}")
        e(f"s_add_u32 s{S + 5}, s{S + 5}, %[lds]")          # staging base
        e(f"v_and_b32 v{T + 5}, 31, v{T}")                  # m
        e(f"v_lshrrev_b32 v{T + 6}, 5, v{T}")               # h
        e(f"v_and_b32 v{T + 7}, 7, v{T + 5}")               # m & 7
        e(f"v_lshl_add_u32 v{T + 8}, v{T + 5}, 8, s{S + 5}")            # base + m * 256
        for q in range(4):
            e(f"v_lshl_add_u32 v{T + 9}, v{T + 6}, 1, {((q >> 1) << 2) | (q & 1)}")      # ((q >> 1) << 2) | (h << 1) | (q & 1)
            e(f"v_xor_b32 v{T + 9}, v{T + 9}, v{T + 7}")
            e(f"v_lshl_add_u32 v{V_STW + q}, v{T + 9}, 4, v{T + 8}")
        # read-back: lane l: row 4 i + (l >> 4), chunk l & 15, stored at (l & 15) ^ ((4 (i & 1) + (l >> 4)) & 7)
        e(f"v_lshrrev_b32 v{T + 5}, 4, v{T}")               # l >> 4
        e(f"v_and_b32 v{T + 6}, 15, v{T}")                  # l & 15
        e(f"v_lshl_add_u32 v{T + 8}, v{T + 5}, 8, s{S + 5}")            # base + (l >> 4) * 256
        for par in range(2):
            e(f"v_add_u32 v{T + 9}, {4 * par}, v{T + 5}")
            e(f"v_xor_b32 v{T + 9}, v{T + 6}, v{T + 9}")
            e(f"v_lshl_add_u32 v{V_STR + par}, v{T + 9}, 4, v{T + 8}")
        # store offset of the lane inside a 4-row group
        e(f"v_mul_lo_u32 v{T + 9}, v{T + 5}, %[ldc]")
        e(f"v_lshl_add_u32 v{V_STO}, v{T + 6}, 4, v{T + 9}")
        e(f"s_lshl_b32 s{S + 6}, s{S + 1}, 7")              # wm * 128 rows
        e(f"s_mul_i32 s{S_ST_WAVE}, s{S + 6}, %[ldc]")
        e(f"s_lshl_b32 s{S + 6}, s{S + 2}, 8")              # wn * 128 columns * 2 B
        e(f"s_add_u32 s{S_ST_WAVE}, s{S_ST_WAVE}, s{S + 6}")
        e(f"s_lshl_b32 s{S_ST_STEP}, %[ldc], 2")
        # buffer descriptor of C: base, stride 0, num_records = 0 (nothing held yet), flags
        e(f"s_mov_b32 s{S_SRD}, %[pc0]")
        e(f"s_and_b32 s{S_SRD + 1}, %[pc1], 0xffff")
        e(f"s_mov_b32 s{S_SRD + 2}, 0")
        e(f"s_mov_b32 s{S_SRD + 3}, 0x00020000")
        e(f"s_mov_b32 s{S_K}, 0")
        e(f"s_mov_b32 s{S_NMY}, %[nmy]")
        if self.epi == "reludrop":
            # chunk index of the lane's first 8 columns in row (lane & 31) relative to the wave's sub-tile (ldc == N: the byte offset / 16), plus
            # the launch's seed term (seed * 0x9E3779B1 + row0 * N / 8)
            e(f"v_and_b32 v{T + 5}, 31, v{T}")
            e(f"v_lshrrev_b32 v{T + 6}, 5, v{T}")
            e(f"v_mul_lo_u32 v{T + 5}, v{T + 5}, %[ldc]")
            e(f"v_lshl_add_u32 v{T + 5}, v{T + 6}, 5, v{T + 5}")
            e(f"v_lshrrev_b32 v{T + 5}, 4, v{T + 5}")
            e(f"v_add_u32 v{V_HL}, %[hseed], v{T + 5}")
            e(f"s_mov_b32 s{S_HK1}, 0x9E3779B1")
            e(f"s_mov_b32 s{S_HK2}, 0x85EBCA77")
            e(f"s_mov_b32 s{S_HXS}, 0x80008000")
            e(f"s_mov_b32 s{S_HPP}, %[hpp]")
            for i in range(4):
                e(f"s_mov_b32 s{S_HC + i}, 0x{0x00EBCA77 + 0x2468 * i:x}")
            e(f"v_mov_b32 v{V_SC2}, %[scale]")
            e(f"v_mov_b32 v{V_SC2 + 1}, %[scale]")
        if self.epi == "dact":
            e(f"v_and_b32 v{T + 5}, 31, v{T}")
            e(f"v_lshrrev_b32 v{T + 6}, 5, v{T}")
            e(f"v_mul_lo_u32 v{T + 5}, v{T + 5}, %[ldc]")
            e(f"v_lshl_add_u32 v{V_ZO}, v{T + 6}, 5, v{T + 5}")
            e(f"s_mov_b32 s{S_ZSRD}, %[pz0]")
            e(f"s_and_b32 s{S_ZSRD + 1}, %[pz1], 0xffff")
            e(f"s_mov_b32 s{S_ZSRD + 2}, %[zbytes]")
            e(f"s_mov_b32 s{S_ZSRD + 3}, 0x00020000")
            e(f"s_mov_b32 s{S_SCALE}, %[scale]")

    def tile_setup(self, sk, store_dst):
        """scalar only: tile ordinal in s{sk} -> running source bases (s36..s39) and the tile's store offset base in s{store_dst}"""
        e = self.e
        X = S_X
        e(f"s_mul_i32 s{X}, s{sk}, %[grid]")
        e(f"s_add_u32 s{X}, s{X}, %[bid]")                  # id = bid + k * grid
        e(f"s_and_b32 s{X + 1}, s{X}, 7")                   # XCD-aware bijective remap (xcd_remap): consecutive tiles stay on one XCD
        e(f"s_lshr_b32 s{X + 2}, s{X}, 3")
        e(f"s_mul_i32 s{X + 3}, s{X + 1}, %[q]")
        e(f"s_min_u32 s{X + 4}, s{X + 1}, %[r]")
        e(f"s_add_u32 s{X + 3}, s{X + 3}, s{X + 4}")
        e(f"s_add_u32 s{X + 3}, s{X + 3}, s{X + 2}")        # logical tile
        # grouped walk: groups of GM tile rows, column by column inside a group (GM = 1: row-major, the walk of the step's shapes, whose W fits
        # the L2; large N: the ~32 tiles an XCD works on at a time cover GM rows x 32 / GM columns instead of one row x 32 columns).
        #   grp = tile / (GM * tilesN), r = tile % (GM * tilesN), gm = rows of this group (GM, or the remainder in the last one),
        #   tn = r / gm, tm = grp * GM + r % gm.   %[walk] = GM | gm_last << 8 | index of a short last group << 16 (0xffff: none)
        e(f"s_mul_hi_u32 s{X + 5}, s{X + 3}, %[magicg]")    # grp
        e(f"s_mul_i32 s{X + 6}, s{X + 5}, %[gsz]")
        e(f"s_sub_u32 s{X + 6}, s{X + 3}, s{X + 6}")        # r
        e(f"s_and_b32 s{X}, %[walk], 0xff")                 # GM
        e(f"s_mul_i32 s{X + 7}, s{X + 5}, s{X}")            # first row of the group
        e(f"s_lshr_b32 s{X + 1}, %[walk], 8")
        e(f"s_and_b32 s{X + 1}, s{X + 1}, 0xff")            # gm_last
        e(f"s_lshr_b32 s{X + 2}, %[walk], 16")              # index of the short group
        e(f"s_cmp_eq_u32 s{X + 5}, s{X + 2}")
        e(f"s_cselect_b32 s{X}, s{X + 1}, s{X}")            # gm
        e(f"s_cselect_b32 s{X + 1}, %[magicl], %[magicm]")  # its division magic
        e(f"s_mul_hi_u32 s{X + 8}, s{X + 6}, s{X + 1}")     # tn = r / gm ...
        e(f"s_cmp_eq_u32 s{X}, 1")
        e(f"s_cselect_b32 s{X + 8}, s{X + 6}, s{X + 8}")    # ... (gm = 1: r itself; 2^32 / 1 has no 32-bit magic)
        e(f"s_mul_i32 s{X + 1}, s{X + 8}, s{X}")
        e(f"s_sub_u32 s{X + 1}, s{X + 6}, s{X + 1}")        # r % gm
        e(f"s_add_u32 s{X + 5}, s{X + 7}, s{X + 1}")        # tm
        e(f"s_mov_b32 s{X + 6}, s{X + 8}")                  # tn
        e(f"s_lshl_b32 s{X + 5}, s{X + 5}, 8")              # m0
        e(f"s_lshl_b32 s{X + 6}, s{X + 6}, 8")              # n0
        # ragged M / N (>= 256): the last tile row / column is shifted up to END at the edge; it overlaps its neighbour, whose rows it
        # recomputes and rewrites with identical values (plain epilogue: no operand is read back)
        e(f"s_min_u32 s{X + 5}, s{X + 5}, %[mlast]")
        e(f"s_min_u32 s{X + 6}, s{X + 6}, %[nlast]")
        e(f"s_mul_i32 s{X + 7}, s{X + 5}, %[lda]")
        e(f"s_mul_hi_u32 s{X + 8}, s{X + 5}, %[lda]")
        e(f"s_add_u32 s{S_PA}, %[pa0], s{X + 7}")
        e(f"s_addc_u32 s{S_PA + 1}, %[pa1], s{X + 8}")
        if not self.tb:
            e(f"s_mul_i32 s{X + 7}, s{X + 6}, %[ldb]")
            e(f"s_mul_hi_u32 s{X + 8}, s{X + 6}, %[ldb]")
            e(f"s_add_u32 s{S_PB}, %[pb0], s{X + 7}")
            e(f"s_addc_u32 s{S_PB + 1}, %[pb1], s{X + 8}")
        else:
            e(f"s_lshl_b32 s{X + 7}, s{X + 6}, 1")
            e(f"s_add_u32 s{S_PB}, %[pb0], s{X + 7}")
            e(f"s_addc_u32 s{S_PB + 1}, %[pb1], 0")
        e(f"s_mul_i32 s{X + 7}, s{X + 5}, %[ldc]")
        e(f"s_lshl_b32 s{X + 8}, s{X + 6}, 1")
        e(f"s_add_u32 s{X + 7}, s{X + 7}, s{X + 8}")
        e(f"s_add_u32 s{store_dst}, s{X + 7}, s{S_ST_WAVE}")

    def advance_p(self):
        return [f"s_add_u32 s{S_PA}, s{S_PA}, s{S_STA}", f"s_addc_u32 s{S_PA + 1}, s{S_PA + 1}, 0",
                f"s_add_u32 s{S_PB}, s{S_PB}, s{S_STB}", f"s_addc_u32 s{S_PB + 1}, s{S_PB + 1}, 0"]

    # ---- write-out pieces
    def conv(self, b):
        e = self.e
        for i in range(8):
            e(f"v_accvgpr_read_b32 v{V_T + 2 * i}, a{16 * b + 2 * i}")
            e(f"v_accvgpr_read_b32 v{V_T + 2 * i + 1}, a{16 * b + 2 * i + 1}")
        if self.epi == "dact":
            assert "z" not in self.vm, "the z tile must have landed long before its conversion"
            for i in range(8):                       # held[8 b + i] still holds z (two bf16): mask = all ones per half iff z > 0
                e(f"v_pk_sub_i16 v{V_ZM + i}, 0, v{V_HELD + 8 * b + i} clamp")
                e(f"v_mul_f32 v{V_T + 2 * i}, s{S_SCALE}, v{V_T + 2 * i}")
                e(f"v_mul_f32 v{V_T + 2 * i + 1}, s{S_SCALE}, v{V_T + 2 * i + 1}")
            for i in range(8):
                e(f"v_pk_ashrrev_i16 v{V_ZM + i}, 15, v{V_ZM + i} op_sel_hi:[0,1]")
                e(f"v_cvt_pk_bf16_f32 v{V_HELD + 8 * b + i}, v{V_T + 2 * i}, v{V_T + 2 * i + 1}")
            for i in range(8):
                e(f"v_and_b32 v{V_HELD + 8 * b + i}, v{V_HELD + 8 * b + i}, v{V_ZM + i}")
            return
        if self.epi == "reludrop":                   # held[8 b + i] holds the keep masks of the pair (hash_stream): out = max(round(acc * scale), 0) & mask
            for i in range(8):
                e(f"v_pk_mul_f32 v[{V_T + 2 * i}:{V_T + 2 * i + 1}], v[{V_T + 2 * i}:{V_T + 2 * i + 1}], v[{V_SC2}:{V_SC2 + 1}]")
            for i in range(8):
                e(f"v_cvt_pk_bf16_f32 v{V_T + 2 * i}, v{V_T + 2 * i}, v{V_T + 2 * i + 1}")
            for i in range(8):
                e(f"v_and_b32 v{V_HELD + 8 * b + i}, v{V_HELD + 8 * b + i}, v{V_T + 2 * i}")
            for i in range(8):
                e(f"v_pk_max_i16 v{V_HELD + 8 * b + i}, v{V_HELD + 8 * b + i}, 0")
            return
        for i in range(8):
            e(f"v_cvt_pk_bf16_f32 v{V_HELD + 8 * b + i}, v{V_T + 2 * i}, v{V_T + 2 * i + 1}")
        if self.epi == "relu":                       # a negative bf16 is a negative int16
            for i in range(8):
                e(f"v_pk_max_i16 v{V_HELD + 8 * b + i}, v{V_HELD + 8 * b + i}, 0")

    def hash_stream(self):
        """(reludrop form) the keep masks of the CURRENT tile, computed into the 128 held registers in the MFMA gaps after the previous tile's
        write-out has left them: v2s_keep8 (v2s_common.h) per aligned chunk of 8 elements -- one 32-bit mix of (seed, chunk index), then per
        element PAIR a rotate + 24-bit multiply = two 16-bit draws, kept iff draw >= p16.  The compare runs on both halves at once: draws and
        threshold are mapped to signed 16-bit (xor 0x8000), sat16((p' - 1) - d') is negative iff d' >= p', its sign spread over the half is
        the AND mask of the bf16.  Entries are (cost, emit): a quarter-rate v_mul_lo_u32 costs 2 units of a gap's budget."""
        e = self.e
        out = [(1, lambda: e(f"s_lshl_b32 s{S_X}, %[ldc], 1")),                      # 32 rows * ldc bytes / 16
               (1, lambda: e(f"s_lshr_b32 s{S_ZROW}, s{S_ST_CUR}, 4"))]
        for bi in range(1, 4):
            out.append((1, lambda bi=bi: e(f"s_add_u32 s{S_ZROW + bi}, s{S_ZROW + bi - 1}, s{S_X}")))
        B, Tm = V_HB, V_HT
        for bi in range(4):
            for bj in range(4):
                for half in range(2):
                    d = V_HELD + (bi * 4 + bj) * 8 + half * 4
                    ops = [(1, f"v_add3_u32 v{B}, v{V_HL}, s{S_ZROW + bi}, {bj * 4 + half}"),
                           (2, f"v_mul_lo_u32 v{B}, v{B}, s{S_HK1}"), (1, f"v_lshrrev_b32 v{Tm}, 15, v{B}"), (1, f"v_xor_b32 v{B}, v{B}, v{Tm}"),
                           (2, f"v_mul_lo_u32 v{B}, v{B}, s{S_HK2}"), (1, f"v_lshrrev_b32 v{Tm}, 13, v{B}"), (1, f"v_xor_b32 v{B}, v{B}, v{Tm}")]
                    for j in range(4):
                        if j:
                            ops.append((1, f"v_alignbit_b32 v{d + j}, v{B}, v{B}, {8 * j}"))
                        ops.append((1, f"v_mul_u32_u24 v{d + j}, v{d + j if j else B}, s{S_HC + j}"))
                        ops.append((1, f"v_xor_b32 v{d + j}, s{S_HXS}, v{d + j}"))
                        ops.append((1, f"v_pk_sub_i16 v{d + j}, s{S_HPP}, v{d + j} clamp"))
                        ops.append((1, f"v_pk_ashrrev_i16 v{d + j}, 15, v{d + j} op_sel_hi:[0,1]"))
                    out.extend((w, (lambda t=t: e(t))) for w, t in ops)
        return out

    def zload_stream(self):
        """(dact form) 32 loads of the current tile's z into the held registers, one per MFMA gap of the Z iteration's first stage: issued
        that early they are OLDER than the DMA requests the iteration's last wait retires, so the queue is back to its steady shape when the
        loop iterations that follow begin"""
        out = []
        for bi in range(4):
            for bj in range(4):
                for half in range(2):
                    reg = V_HELD + (bi * 4 + bj) * 8 + half * 4
                    out.append(lambda reg=reg, bi=bi, bj=bj, half=half: self.vm_op(
                        "z", f"buffer_load_dwordx4 v[{reg}:{reg + 3}], v{V_ZO}, s[{S_ZSRD}:{S_ZSRD + 3}], s{S_ZROW + bi} offen offset:{bj * 64 + half * 16}"))
        return out

    def wops(self, c):
        out = []
        for j in range(8):
            bj, half = j >> 1, j & 1
            q = (bj & 1) * 2 + half
            reg = V_HELD + (c * 4 + bj) * 8 + half * 4
            out.append(lambda reg=reg, q=q, bj=bj, c=c: self.lds_op(f"W{c}", f"ds_write_b128 v{V_STW + q}, v[{reg}:{reg + 3}] offset:{(bj >> 1) * 128}"))
        return out

    def rops(self, c):
        out = []
        for i in range(8):
            reg = V_HELD + c * 32 + 4 * i
            out.append(lambda reg=reg, i=i, c=c: self.lds_op(f"R{c}_{i}", f"ds_read_b128 v[{reg}:{reg + 3}], v{V_STR + (i & 1)} offset:{i * 1024}"))
        return out

    def sops(self, c):
        out = []
        for i in range(8):
            reg = V_HELD + c * 32 + 4 * i

            def st(reg=reg, i=i, c=c):
                self.need({f"R{c}_{i}"})
                self.vm_op("st", f"buffer_store_dwordx4 v[{reg}:{reg + 3}], v{V_STO}, s[{S_SRD}:{S_SRD + 3}], s{S_ST_PREV} offen{STORE_MOD}")
            out.append(st)
            out.append(lambda: self.e(f"s_add_u32 s{S_ST_PREV}, s{S_ST_PREV}, s{S_ST_STEP}"))
        return out

    def writeout_stream(self):
        nop = lambda: None
        seq = self.wops(0) + self.rops(0) + self.wops(1) + self.sops(0) + self.rops(1) + self.wops(2) + self.sops(1) + self.rops(2) + \
            self.wops(3) + self.sops(2) + self.rops(3) + [nop] * 6 + self.sops(3)
        return seq

    def body_p(self, kind):
        """4 stages; kind: 'T' = first iteration of a tile (conversion of the previous tile in step 0, write-out stream starts),
        'S' = second (the stream runs out), 'P' = any later one (with the tile switch of the DMA stream when it is the tile's last)"""
        for st in range(4):
            gaps, pre = {}, {}
            self.fill_even(st, gaps)
            first = kind == "T" and st == 0
            if kind == "Z" and st == 0:
                # z offsets of the wave's four block rows of the CURRENT tile (= its store offsets: ldz == ldc), then the load stream
                self.e(f"s_lshl_b32 s{S_X}, %[ldc], 5")
                self.e(f"s_mov_b32 s{S_ZROW}, s{S_ST_CUR}")
                for bi in range(1, 4):
                    self.e(f"s_add_u32 s{S_ZROW + bi}, s{S_ZROW + bi - 1}, s{S_X}")
                self.bg = self.zload_stream()
                if ZSPREAD > 1:
                    self.bg = [x for ld in self.bg for x in [ld] + [lambda: None] * (ZSPREAD - 1)]
            if first:
                for b in range(16):
                    pre[b] = [lambda b=b: self.conv(b)]
            rate = self.bg_rate
            if first:
                self.bg_rate = 0                             # the stream starts after the conversion step
            self.step(0, gaps, first=first, pre=pre)
            self.bg_rate = rate
            if first:
                self.bg = self.writeout_stream()
                if self.epi == "reludrop":
                    self.bg = self.bg + self.hash_stream()
            gaps = {}
            self.fill_odd(st, gaps, self.advance_p(), extra_sync=self.tile_switch if (kind == "P" and st == 0) else None)
            self.step(1, gaps)
            self.stage_ctr += 1
        assert not self.bg or kind == "T" or (self.epi == "reludrop" and kind in ("S", "H")) or (kind == "Z" and ZSPREAD > 1), "background stream did not finish"

    def tile_switch(self):
        """in the LAST iteration of a tile, after the tile's last stage has been requested: point the DMA stream at the block's next tile
        (or at the same tile again when there is none)"""
        e = self.e
        e(f"s_cmp_lg_u32 s{S_IT}, 1")
        e("s_cbranch_scc1 L_a4p_nosw_%=")
        e(f"s_add_u32 s{S_X + 9}, s{S_K}, 1")
        e(f"s_sub_u32 s{S_X + 10}, s{S_NMY}, 1")
        e(f"s_min_u32 s{S_X + 9}, s{S_X + 9}, s{S_X + 10}")
        self.tile_setup(S_X + 9, S_ST_NEXT)
        e("L_a4p_nosw_%=:")

    def main_p(self):
        e = self.e
        self.setup_p()
        self.tile_setup(S_K, S_ST_CUR)
        e(f"s_mov_b32 s{S_ST_PREV}, s{S_ST_CUR}")
        e("s_nop 4")
        self.prologue(self.advance_p)
        entry = (list(self.lgkm), list(self.vm))
        shape = lambda: (list(self.lgkm), ["st" if t == "st" else "d" for t in self.vm])
        entry_shape = shape()
        e("L_a4p_tile_%=:")
        self.body_p("T")
        self.body_p("S")
        assert shape() == entry_shape, (shape(), entry_shape)
        if self.epi == "dact":
            self.body_p("Z")
            if ZSPREAD > 1:
                self.body_p("Z2")
                assert not self.bg
            assert "z" not in self.vm and (list(self.lgkm), ["d"] * len(self.vm)) == (entry_shape[0], ["d"] * 20), (self.lgkm, self.vm)
        if self.epi == "reludrop":                   # the hash stream needs the gaps of two more iterations (K >= 640)
            self.body_p("H")
            self.body_p("H")
            assert not self.bg, f"hash stream did not finish: {len(self.bg)} entries left"
            assert shape() == entry_shape
        e(f"s_sub_u32 s{S_IT}, %[niter], {(4 if ZSPREAD > 1 else 3) if self.epi == 'dact' else (4 if self.epi == 'reludrop' else 2)}")
        e("L_a4p_loop_%=:")
        self.vm = [f"stage{self.stage_ctr + 1}"] * 8 + [f"stage{self.stage_ctr + 2}"] * 8 + [f"stage{self.stage_ctr + 3}"] * 4
        self.body_p("P")
        assert shape() == entry_shape
        e(f"s_sub_u32 s{S_IT}, s{S_IT}, 1")
        e(f"s_cmp_lg_u32 s{S_IT}, 0")
        e("s_cbranch_scc1 L_a4p_loop_%=")
        # tile boundary
        e(f"s_mov_b32 s{S_ST_PREV}, s{S_ST_CUR}")
        e(f"s_mov_b32 s{S_ST_CUR}, s{S_ST_NEXT}")
        e(f"s_mov_b32 s{S_SRD + 2}, %[cbytes]")
        e(f"s_add_u32 s{S_K}, s{S_K}, 1")
        e(f"s_cmp_lt_u32 s{S_K}, s{S_NMY}")
        e("s_cbranch_scc1 L_a4p_tile_%=")
        # drain: the block's last tile is still in the accumulators
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.vm, self.lgkm = [], []
        for b in range(16):
            self.conv(b)
        for f in self.writeout_stream():
            f()
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        return self.lines


def clobbers_p():
    c = ['"memory"', '"scc"', '"vcc"']
    c += [f'"v{i}"' for i in range(NV_CLOBBER_P)]
    c += [f'"a{i}"' for i in range(256)]
    c += [f'"s{i}"' for i in range(36, NS_CLOBBER_P)]
    return ", ".join(c)


def dump_lines(bi):
    """accumulator blocks (bi, 0..3) -> fp32 staging: lane (h, m): row wm * 32 + m of the pass, columns wn * 128 + 32 bj + 16 h + r;
    %[sa] = this lane's staging byte address (row * 1040 + (wn * 128 + 16 h) * 4)"""
    out = []
    for bj in range(4):
        for q in range(4):
            acc = 16 * (4 * bi + bj) + 4 * q
            out.append(f"ds_write_b128 %[sa], a[{acc}:{acc + 3}] offset:{bj * 128 + q * 16}")
    out.append("s_waitcnt lgkmcnt(0)")
    return out


def clobbers():
    c = ['"memory"', '"scc"', '"vcc"']
    c += [f'"v{i}"' for i in range(NV_CLOBBER)]
    c += [f'"a{i}"' for i in range(256)]
    c += [f'"s{i}"' for i in range(36, 56)]
    return ", ".join(c)


def as_c_string(lines):
    return "\n".join('  "' + ln + '\\n\\t"' for ln in lines)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "v2s_gemm_a4.inc"
    parts = ["// GENERATED by gen_gemm_a4.py -- do not edit.  The hand-scheduled K loop of gemm_a4_kernel (see the generator's header).",
             "// clang-format off"]
    stats = {}
    for tb, ta in ((False, False), (True, False), (True, True)):
        g = Gen(tb, ta)
        lines = g.main()
        stats[tb] = (len(lines), g.nmfma)
        parts.append(f"#define A4_MAIN_{'TN' if ta else ('NN' if tb else 'NT')} \\")
        parts.append(" \\\n".join('  "' + ln + '\\n\\t"' for ln in lines))
        parts.append("")
    parts.append("#ifdef V2S_A4_ABLATIONS")
    for abl in ("nodma", "noread", "none", "nobar", "nowait"):
        g = Gen(False, False, abl)
        parts.append(f"#define A4_MAIN_NT_{abl.upper()} \\")
        parts.append(" \\\n".join('  "' + ln + '\\n\\t"' for ln in g.main()))
        parts.append("")
    parts.append("#endif")
    for tb in (False, True):
        g = GenP(tb)
        lines = g.main_p()
        stats[("p", tb)] = (len(lines), g.nmfma)
        parts.append(f"#define A4P_MAIN_{'NN' if tb else 'NT'} \\")
        parts.append(" \\\n".join('  "' + ln + '\\n\\t"' for ln in lines))
        parts.append("")
    g = GenP(True, "dact")
    lines = g.main_p()
    stats[("p", "dact")] = (len(lines), g.nmfma)
    parts.append("#define A4P_MAIN_NN_DACT \\")
    parts.append(" \\\n".join('  "' + ln + '\\n\\t"' for ln in lines))
    parts.append("")
    for epi in ("relu", "reludrop"):
        g = GenP(False, epi)
        lines = g.main_p()
        stats[("p", epi)] = (len(lines), g.nmfma)
        parts.append(f"#define A4P_MAIN_NT_{epi.upper()} \\")
        parts.append(" \\\n".join('  "' + ln + '\\n\\t"' for ln in lines))
        parts.append("")
    parts.append(f"#define A4P_CLOBBERS {clobbers_p()}")
    for bi in range(4):
        parts.append(f"#define A4_DUMP_{bi} \\")
        parts.append(" \\\n".join('  "' + ln + '\\n\\t"' for ln in dump_lines(bi)))
        parts.append("")
    parts.append(f"#define A4_CLOBBERS {clobbers()}")
    parts.append("// clang-format on")
    with open(out, "w") as f:
        f.write("\n".join(parts) + "\n")
    print(f"wrote {out}: NT {stats[False][0]} lines / {stats[False][1]} MFMAs, NN {stats[True][0]} lines / {stats[True][1]} MFMAs; "
          f"persistent NT {stats[('p', False)][0]} lines, NN {stats[('p', True)][0]} lines")


if __name__ == "__main__":
    main()
