#!/usr/bin/env python3
"""Functional emulator of the instruction subset gen_gemm_a4.py emits (one workgroup = 4 waves of 64 lanes), used to check the
hand-scheduled K loop of gemm_a4_kernel WITHOUT a GPU: address arithmetic, LDS images and swizzles, fragment <-> MFMA mapping, literal
register allocation, loop control -- and, through adversarial completion semantics, every counted s_waitcnt and the barrier protocol:

  * a ds_read's destination registers are POISONED at issue; the data is sampled from LDS and written when an s_waitcnt lgkmcnt(N)
    retires the read ("lazy") or right at issue ("eager");
  * an LDS-DMA (global_load_lds) writes LDS when an s_waitcnt vmcnt(N) of the issuing wave retires it ("lazy": a read that was not
    ordered behind wait + barrier sees stale bytes) or at issue ("eager": a request that was issued before every wave finished reading
    the slot destroys the data);
  * waves are interleaved at instruction granularity by a seeded random scheduler (or run barrier to barrier in a fixed order).
A schedule is accepted only if the result is right under every combination.

This is test infrastructure (tests/test_gemm_a4_emu.py); the semantics follow the CDNA3/4 ISA as used by the existing kernels of this
library (ds_read_b64_tr_b16 lane mapping: v2s_gemm.hip read_frag; LDS-DMA: M0 + 16 * lane).
usage: python tools/a4_emu.py [--nn] [M N K]"""
import re
import sys

import numpy as np

POISON = np.uint32(0x7FC0DEAD)


def parse_inc(path):
    """macro name -> list of instruction strings"""
    macros, cur, name = {}, None, None
    for ln in open(path):
        m = re.match(r"#define (\w+) \\", ln)
        if m:
            name, cur = m.group(1), []
            macros[name] = cur
            continue
        if cur is not None:
            m = re.match(r'\s*"(.*?)\\n\\t"', ln)
            if m:
                cur.append(m.group(1))
            elif not ln.strip():
                cur = None
    return macros


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


class Wave:
    def __init__(self, wid, blk):
        self.wid, self.blk = wid, blk
        self.s = np.zeros(128, np.uint32)
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.uint32)
        self.scc = 0
        self.m0 = np.uint32(0)
        self.pc = 0
        self.lgkm = []       # pending LDS ops (closures), oldest first
        self.vm = []
        self.at_barrier = False
        self.done = False
        self.nmfma = 0


class Block:
    def __init__(self, prog, gmem, lazy_ds, lazy_dma, lds_bytes=163840):
        self.prog = prog
        self.labels = {}
        for i, ins in enumerate(prog):
            if ins.endswith(":"):
                self.labels[ins[:-1]] = i
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.gmem = gmem           # list of (base, np.uint8 array)
        self.lazy_ds, self.lazy_dma = lazy_ds, lazy_dma
        self.waves = [Wave(w, self) for w in range(4)]

    # ---- memory
    def gread(self, addr, n):
        for base, arr in self.gmem:
            if base <= addr and addr + n <= base + len(arr):
                return arr[addr - base: addr - base + n]
        raise RuntimeError(f"global read out of bounds: 0x{addr:x} (+{n})")

    def gwrite(self, addr, data):
        for base, arr in self.gmem:
            if base <= addr and addr + len(data) <= base + len(arr):
                arr[addr - base: addr - base + len(data)] = data
                return
        raise RuntimeError(f"global write out of bounds: 0x{addr:x} (+{len(data)})")

    # ---- operand decoding
    def src(self, w, tok, vec=True):
        """value of a source operand: scalar -> np.uint32, vector -> np.uint32[64]"""
        tok = tok.strip()
        if re.fullmatch(r"v\d+", tok):
            return w.v[int(tok[1:])]
        if re.fullmatch(r"s\d+", tok):
            return np.uint32(w.s[int(tok[1:])])
        if tok == "m0":
            return np.uint32(w.m0)
        if re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", tok):
            return np.uint32(int(tok, 0) & 0xFFFFFFFF)
        raise RuntimeError(f"bad source operand {tok!r}")

    @staticmethod
    def rng(tok):
        m = re.fullmatch(r"([vsa])\[(\d+):(\d+)\]", tok.strip())
        if m:
            return m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
        m = re.fullmatch(r"([vsa])(\d+)", tok.strip())
        return m.group(1), int(m.group(2)), 1

    def setv(self, w, tok, val):
        k, i, n = self.rng(tok)
        assert k == "v" and n == 1, tok
        w.v[i] = np.broadcast_to(np.asarray(val, np.uint32), (64,)).copy()

    def sets(self, w, tok, val):
        tok = tok.strip()
        if tok == "m0":
            w.m0 = np.uint32(val)
            return
        k, i, n = self.rng(tok)
        assert k == "s" and n == 1, tok
        w.s[i] = np.uint32(int(val) & 0xFFFFFFFF)

    # ---- waits
    def retire(self, q, n):
        while len(q) > n:
            q.pop(0)()

    # ---- one instruction
    def step(self, w):
        ins = self.prog[w.pc]
        w.pc += 1
        if ins.endswith(":") or not ins:
            return
        m = re.match(r"(\S+)\s*(.*)", ins)
        op, rest = m.group(1), m.group(2)
        offset = 0
        mo = re.search(r"offset:(\d+)", rest)
        if mo:
            offset = int(mo.group(1))
            rest = rest[:mo.start()].strip()
        ops = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
        S, V = (lambda t: self.src(w, t)), (lambda t: np.broadcast_to(self.src(w, t), (64,)))
        u64 = lambda x: np.asarray(x, np.uint64)

        if op == "s_nop" or op == "s_setprio" or op == "s_sleep":
            return
        if op == "s_mov_b32":
            self.sets(w, ops[0], S(ops[1])); return
        if op in ("s_add_u32", "s_addc_u32", "s_sub_u32"):
            a, b = int(S(ops[1])), int(S(ops[2]))
            if op == "s_add_u32":
                r = a + b; w.scc = int(r >> 32 != 0)
            elif op == "s_addc_u32":
                r = a + b + w.scc; w.scc = int(r >> 32 != 0)
            else:
                r = a - b; w.scc = int(b > a)
            self.sets(w, ops[0], r & 0xFFFFFFFF); return
        if op in ("s_mul_i32", "s_mul_hi_u32", "s_min_u32", "s_max_u32"):
            a, b = int(S(ops[1])), int(S(ops[2]))
            r = {"s_mul_i32": (a * b) & 0xFFFFFFFF, "s_mul_hi_u32": (a * b) >> 32, "s_min_u32": min(a, b), "s_max_u32": max(a, b)}[op]
            if op in ("s_min_u32", "s_max_u32"):
                w.scc = int(r == a)
            self.sets(w, ops[0], r); return
        if op in ("s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32"):
            a, b = int(S(ops[1])), int(S(ops[2]))
            r = {"s_lshl_b32": (a << (b & 31)), "s_lshr_b32": a >> (b & 31), "s_and_b32": a & b, "s_or_b32": a | b}[op] & 0xFFFFFFFF
            w.scc = int(r != 0)
            self.sets(w, ops[0], r); return
        if op in ("s_cmp_lg_u32", "s_cmp_eq_u32", "s_cmp_lt_u32", "s_cmp_ge_u32"):
            a, b = int(S(ops[0])), int(S(ops[1]))
            w.scc = int({"s_cmp_lg_u32": a != b, "s_cmp_eq_u32": a == b, "s_cmp_lt_u32": a < b, "s_cmp_ge_u32": a >= b}[op]); return
        if op == "s_cselect_b32":
            self.sets(w, ops[0], S(ops[1]) if w.scc else S(ops[2])); return
        if op in ("s_cbranch_scc1", "s_cbranch_scc0", "s_branch"):
            take = op == "s_branch" or (w.scc == 1) == (op == "s_cbranch_scc1")
            if take:
                w.pc = self.labels[ops[0]]
            return
        if op == "s_waitcnt":
            mv = re.search(r"vmcnt\((\d+)\)", ins)
            ml = re.search(r"lgkmcnt\((\d+)\)", ins)
            if mv:
                self.retire(w.vm, int(mv.group(1)))
            if ml:
                self.retire(w.lgkm, int(ml.group(1)))
            return
        if op == "s_barrier":
            w.at_barrier = True; return
        if op == "v_readfirstlane_b32":
            self.sets(w, ops[0], V(ops[1])[0]); return
        if op in ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_min_u32", "v_max_u32", "v_mul_lo_u32",
                  "v_lshlrev_b32", "v_lshrrev_b32"):
            a, b = u64(V(ops[1])), u64(V(ops[2]))
            r = {"v_and_b32": lambda: a & b, "v_or_b32": lambda: a | b, "v_xor_b32": lambda: a ^ b, "v_add_u32": lambda: a + b,
                 "v_sub_u32": lambda: a - b, "v_min_u32": lambda: np.minimum(a, b), "v_max_u32": lambda: np.maximum(a, b),
                 "v_mul_lo_u32": lambda: a * b, "v_lshlrev_b32": lambda: b << (a & np.uint64(31)),
                 "v_lshrrev_b32": lambda: b >> (a & np.uint64(31))}[op]()
            self.setv(w, ops[0], (r & np.uint64(0xFFFFFFFF)).astype(np.uint32)); return
        if op == "v_add3_u32":
            r = u64(V(ops[1])) + u64(V(ops[2])) + u64(V(ops[3]))
            self.setv(w, ops[0], (r & np.uint64(0xFFFFFFFF)).astype(np.uint32)); return
        if op == "v_alignbit_b32":                             # D = ({S0, S1} >> S2[4:0]) & 0xffffffff
            sh = u64(V(ops[3])) & np.uint64(31)
            r = ((u64(V(ops[1])) << np.uint64(32)) | u64(V(ops[2]))) >> sh
            self.setv(w, ops[0], (r & np.uint64(0xFFFFFFFF)).astype(np.uint32)); return
        if op == "v_mul_u32_u24":
            r = (u64(V(ops[1])) & np.uint64(0xFFFFFF)) * (u64(V(ops[2])) & np.uint64(0xFFFFFF))
            self.setv(w, ops[0], (r & np.uint64(0xFFFFFFFF)).astype(np.uint32)); return
        if op == "v_pk_mul_f32":
            (kd, d0, dn), (ka, a0, an), (kb, b0, bn) = self.rng(ops[0]), self.rng(ops[1]), self.rng(ops[2])
            assert dn == an == bn == 2 and kd == ka == kb == "v", ins
            with np.errstate(all="ignore"):
                lo = (w.v[a0].view(np.float32) * w.v[b0].view(np.float32)).astype(np.float32).view(np.uint32)
                hi = (w.v[a0 + 1].view(np.float32) * w.v[b0 + 1].view(np.float32)).astype(np.float32).view(np.uint32)
            w.v[d0], w.v[d0 + 1] = lo.copy(), hi.copy(); return
        if op == "v_lshl_add_u32":
            r = (u64(V(ops[1])) << (u64(V(ops[2])) & np.uint64(31))) + u64(V(ops[3]))
            self.setv(w, ops[0], (r & np.uint64(0xFFFFFFFF)).astype(np.uint32)); return
        if op == "v_mov_b32":
            self.setv(w, ops[0], V(ops[1])); return
        if op == "v_accvgpr_read_b32":
            k, i, n = self.rng(ops[1]); self.setv(w, ops[0], w.a[i]); return
        if op == "v_cvt_pk_bf16_f32":
            def rne(x):
                x = x.astype(np.uint64)
                return (((x + np.uint64(0x7FFF) + ((x >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.uint32)
            self.setv(w, ops[0], rne(V(ops[1])) | (rne(V(ops[2])) << np.uint32(16))); return
        if op == "v_mfma_f32_32x32x16_bf16":
            kd, d0, dn = self.rng(ops[0]); ka, a0, an = self.rng(ops[1]); kb, b0, bn = self.rng(ops[2])
            assert kd == "a" and dn == 16 and an == 4 and bn == 4 and ka == "v" and kb == "v"
            # srcA: lane l holds row i = l & 31, k = 8 (l >> 5) + j, two bf16 per register; srcB: column j = l & 31, same k map
            def unpack(base):
                regs = w.v[base:base + 4]                                    # [4][64]
                lo = bf16_to_f32((regs & np.uint32(0xFFFF)).astype(np.uint16))
                hi = bf16_to_f32((regs >> np.uint32(16)).astype(np.uint16))
                per_lane = np.stack([lo, hi], 1).reshape(8, 64)              # element j = 2 reg + half
                mat = np.zeros((32, 16), np.float32)
                for l in range(64):
                    mat[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = per_lane[:, l]
                return mat
            A, B = unpack(a0), unpack(b0)                                    # [32 rows][16 k] each
            if ops[3].strip() == "0":
                C = np.zeros((32, 32), np.float32)
            else:
                kc, c0, cn = self.rng(ops[3]); assert kc == "a" and cn == 16
                C = self.acc_to_mat(w, c0)
            D = C + A.astype(np.float64) @ B.astype(np.float64).T            # D[i][j] = sum_k A[i][k] B[j][k]
            self.mat_to_acc(w, d0, D.astype(np.float32))
            w.nmfma += 1
            return
        if op == "ds_read_b128" or op == "ds_read_b64_tr_b16":
            kd, d0, dn = self.rng(ops[0])
            addr = V(ops[1]).astype(np.int64) + offset
            arr = w.v if kd == "v" else w.a
            arr[d0:d0 + dn] = POISON
            tr = op != "ds_read_b128"
            box = {}

            def sample(addr=addr.copy(), dn=dn, tr=tr, box=box):
                """read LDS (once): at the latest moment the ordering rules allow -- retirement, or just before a later LDS write of the SAME
                wave (a wave's LDS operations execute in order)"""
                if "d" in box:
                    return
                out = np.zeros((dn, 64), np.uint32)
                if not tr:
                    for l in range(64):
                        out[:, l] = self.lds[addr[l]:addr[l] + 16].view(np.uint32)
                else:
                    # 16-lane groups: lane j supplies the address of 4 bf16 (k-row j >> 2 of the block, columns 4 (j & 3) ..); lane i receives
                    # column i: for kk = 0..3 the element (i & 3) of the run supplied by lane 4 kk + (i >> 2)
                    for g in range(4):
                        for i in range(16):
                            vals = []
                            for kk in range(4):
                                srcl = g * 16 + 4 * kk + (i >> 2)
                                a_ = addr[srcl] + 2 * (i & 3)
                                vals.append(int(self.lds[a_]) | (int(self.lds[a_ + 1]) << 8))
                            out[0, g * 16 + i] = vals[0] | (vals[1] << 16)
                            out[1, g * 16 + i] = vals[2] | (vals[3] << 16)
                box["d"] = out

            def complete(d0=d0, dn=dn, arr=arr, box=box, sample=sample):
                sample()
                arr[d0:d0 + dn] = box["d"]
            complete.sample = sample
            if self.lazy_ds:
                assert len(w.lgkm) < 15
                w.lgkm.append(complete)
            else:
                complete(); w.lgkm.append(lambda: None)
            return
        if op == "ds_write_b128":
            kd, d0, dn = self.rng(ops[1])
            addr = V(ops[0]).astype(np.int64) + offset
            arr = w.v if kd == "v" else w.a
            data = arr[d0:d0 + 4].copy()
            for f in w.lgkm:                                 # older reads of this wave execute first
                if hasattr(f, "sample"):
                    f.sample()
            for l in range(64):
                self.lds[addr[l]:addr[l] + 16] = np.ascontiguousarray(data[:, l]).view(np.uint8)
            w.lgkm.append(lambda: None)               # (a dump statement issues 16 writes before its lgkmcnt(0): the hardware stalls at 15)
            return
        if op == "global_load_lds_dwordx4":
            voff = V(ops[0]).astype(np.int64)
            k, sb, n = self.rng(ops[1]); assert n == 2
            base = int(w.s[sb]) | (int(w.s[sb + 1]) << 32)
            m0 = int(w.m0)
            srcs = [base + int(voff[l]) + offset for l in range(64)]

            def complete(srcs=srcs, m0=m0):
                for l in range(64):
                    self.lds[m0 + 16 * l: m0 + 16 * l + 16] = self.gread(srcs[l], 16)
            if self.lazy_dma:
                w.vm.append(complete)
            else:
                complete(); w.vm.append(lambda: None)
            assert len(w.vm) <= 63
            return
        if op == "v_mul_f32":
            a, b = V(ops[1]).view(np.float32), V(ops[2]).view(np.float32)
            with np.errstate(all="ignore"):
                self.setv(w, ops[0], (a * b).astype(np.float32).view(np.uint32))
            return
        if op in ("v_pk_sub_i16", "v_pk_ashrrev_i16", "v_pk_max_i16"):
            clamp = bool(re.search(r"\bclamp\b", rest))
            opsel = "op_sel_hi:[0,1]" in rest
            clean = re.sub(r"op_sel_hi:\[[^\]]*\]|\bclamp\b", "", rest).strip()
            toks = [t.strip() for t in clean.split(",")]

            def halves(t, inline_lo_for_hi=False):
                x = V(t).astype(np.uint32)
                lo = (x & np.uint32(0xFFFF)).astype(np.uint16).view(np.int16).astype(np.int64)
                hi = (x >> np.uint32(16)).astype(np.uint16).view(np.int16).astype(np.int64)
                return lo, (lo if inline_lo_for_hi else hi)
            inline1 = re.fullmatch(r"-?\d+", toks[1]) is not None
            a_lo, a_hi = halves(toks[1], inline_lo_for_hi=(opsel and inline1))
            b_lo, b_hi = halves(toks[2])
            if op == "v_pk_sub_i16":
                r_lo, r_hi = a_lo - b_lo, a_hi - b_hi
                if clamp:
                    r_lo, r_hi = np.clip(r_lo, -32768, 32767), np.clip(r_hi, -32768, 32767)
            elif op == "v_pk_max_i16":
                r_lo, r_hi = np.maximum(a_lo, b_lo), np.maximum(a_hi, b_hi)
            else:                                              # D = S1 >> S0 (arithmetic), per half
                r_lo, r_hi = b_lo >> (a_lo & 15), b_hi >> (a_hi & 15)
            r = (r_lo.astype(np.int16).view(np.uint16).astype(np.uint32)) | (r_hi.astype(np.int16).view(np.uint16).astype(np.uint32) << np.uint32(16))
            self.setv(w, toks[0], r)
            return
        if op == "buffer_load_dwordx4":
            # buffer_load_dwordx4 vdst[4], voffset, srd[4], soffset offen [offset:N]   (out of range -> zeros)
            kd, d0, dn = self.rng(ops[0]); assert dn == 4 and kd == "v"
            voff = V(ops[1]).astype(np.int64)
            k, sb, n = self.rng(ops[2]); assert n == 4
            base = int(w.s[sb]) | ((int(w.s[sb + 1]) & 0xFFFF) << 32)
            nrec = int(w.s[sb + 2])
            soff = int(S(ops[3].replace("offen", "").strip()))
            w.v[d0:d0 + 4] = POISON
            addrs = [(base + soff + int(voff[l]) + offset) if int(voff[l]) + offset + 16 <= nrec else None for l in range(64)]

            def complete(d0=d0, addrs=addrs):
                for l in range(64):
                    w.v[d0:d0 + 4, l] = 0 if addrs[l] is None else self.gread(addrs[l], 16).view(np.uint32)
            if self.lazy_dma:
                w.vm.append(complete)
            else:
                complete(); w.vm.append(lambda: None)
            assert len(w.vm) <= 63
            return
        if op == "buffer_store_dwordx4":
            # buffer_store_dwordx4 vdata[4], voffset, srd[4], soffset offen   (raw buffer: out of range when voffset + 16 > num_records)
            ops[3] = re.sub(r"\s+(sc0|sc1|nt)\b", "", ops[3])           # cache-policy modifiers: no functional effect
            ops[3] = re.sub(r"\s+(sc0|sc1|nt)\b", "", ops[3])
            assert ops[3].endswith("offen"), ins
            kd, d0, dn = self.rng(ops[0]); assert dn == 4
            voff = V(ops[1]).astype(np.int64)
            k, sb, n = self.rng(ops[2]); assert n == 4
            base = int(w.s[sb]) | ((int(w.s[sb + 1]) & 0xFFFF) << 32)
            nrec = int(w.s[sb + 2])
            soff = int(S(ops[3].replace("offen", "").strip()))
            data = (w.v if kd == "v" else w.a)[d0:d0 + 4].copy()
            for l in range(64):
                if int(voff[l]) + 16 <= nrec:
                    self.gwrite(base + soff + int(voff[l]), np.ascontiguousarray(data[:, l]).view(np.uint8))
            w.vm.append(lambda: None)
            assert len(w.vm) <= 63
            return
        raise RuntimeError(f"unhandled instruction: {ins}")

    # accumulator block <-> 32x32 matrix D[i][j]: lane l: j = l & 31, i = 8 (r >> 2) + 4 (l >> 5) + (r & 3)
    _R, _L = np.meshgrid(np.arange(16), np.arange(64), indexing="ij")
    _I, _J = 8 * (_R >> 2) + 4 * (_L >> 5) + (_R & 3), _L & 31

    def acc_to_mat(self, w, base):
        D = np.zeros((32, 32), np.float32)
        D[self._I, self._J] = w.a[base:base + 16].view(np.float32)
        return D

    def mat_to_acc(self, w, base, D):
        w.a[base:base + 16] = np.ascontiguousarray(D[self._I, self._J]).view(np.uint32)

    # ---- run the workgroup
    def run(self, sched="random", seed=0, max_steps=10_000_000):
        rs = np.random.RandomState(seed)
        n = len(self.prog)
        steps = 0
        while True:
            live = [w for w in self.waves if not w.done]
            if not live:
                return
            if all(w.at_barrier for w in live):
                assert len(live) == 4, "a wave ended while others wait at a barrier"
                for w in live:
                    w.at_barrier = False
                continue
            runnable = [w for w in live if not w.at_barrier]
            if sched == "random":
                w = runnable[rs.randint(len(runnable))]
                burst = rs.randint(1, 40)
            elif sched == "fwd":
                w, burst = runnable[0], 1 << 30
            else:
                w, burst = runnable[-1], 1 << 30
            for _ in range(burst):
                if w.at_barrier or w.done:
                    break
                if w.pc >= n:
                    assert not w.lgkm or True
                    w.done = True
                    break
                self.step(w)
                steps += 1
                assert steps < max_steps, "runaway"


def render(lines, sub):
    out = []
    for ln in lines:
        for k, v in sub.items():
            ln = ln.replace(f"%[{k}]", v)
        out.append(ln.replace("%=", "0"))
    return out


SUB = dict(tid="v120", pa0="s8", pa1="s9", pb0="s10", pb1="s11", lda="s12", ldb="s13", m0="s14", n0="s15", mmax="s16", nmax="s17",
           niter="s18", lds="s19", sa="v121")


def check(inc, tb, M, N, K, tile=(0, 0), seed=1, lazy_ds=True, lazy_dma=True, sched="random", verbose=False, ta=False):
    """run one 256 x 256 tile of C = X . W^T (W given as [N][K], or as [K][N] when tb) through the generated main loop + dumps; returns max
    abs error against the fp64 product of the bf16 inputs"""
    macros = parse_inc(inc)
    rs = np.random.RandomState(seed)
    X = (rs.randn(M, K) * 0.5).astype(np.float32)
    W = (rs.randn(N, K) * 0.5).astype(np.float32)
    to_bf = lambda f: ((f.view(np.uint32) + 0x7FFF + ((f.view(np.uint32) >> 16) & 1)) >> 16).astype(np.uint16)
    Xb, Wb = to_bf(X), to_bf(W)
    Xf, Wf = bf16_to_f32(Xb).astype(np.float64), bf16_to_f32(Wb).astype(np.float64)
    ref = Xf @ Wf.T
    if ta:
        Amem, lda = np.ascontiguousarray(Xb.T), M            # [K][M] (weight gradients: dY^T)
        assert tb
    else:
        Amem, lda = Xb, K
    if tb:
        Bmem, ldb = np.ascontiguousarray(Wb.T), N            # [K][N]
    else:
        Bmem, ldb = Wb, K
    PA, PB = 0x10000000, 0x30000000
    gmem = [(PA, Amem.view(np.uint8).reshape(-1)), (PB, Bmem.view(np.uint8).reshape(-1))]
    blk = Block(render(macros["A4_MAIN_TN" if ta else ("A4_MAIN_NN" if tb else "A4_MAIN_NT")], SUB), gmem, lazy_ds, lazy_dma)
    tm, tn = tile
    m0, n0 = tm * 256, tn * 256
    for w in blk.waves:
        w.v[120] = np.arange(64, dtype=np.uint32) + 64 * w.wid
        w.s[8], w.s[9] = PA & 0xFFFFFFFF, PA >> 32
        w.s[10], w.s[11] = PB & 0xFFFFFFFF, PB >> 32
        w.s[12], w.s[13] = lda * 2, ldb * 2
        w.s[14], w.s[15] = m0, n0
        w.s[16] = (((M + 7) & ~7) - 8) if ta else M - 1
        w.s[17] = (((N + 7) & ~7) - 8) if tb else N - 1
        w.s[18] = K // 128
        w.s[19] = 0
    blk.run(sched=sched, seed=seed)
    assert all(not w.vm and not w.lgkm for w in blk.waves), "operations outstanding at the end of the main loop"
    assert all(w.nmfma == 16 * (K // 16) for w in blk.waves), [w.nmfma for w in blk.waves]
    # epilogue dumps: pass bi -> fp32 staging [64][260]
    got = np.full((256, 256), np.nan, np.float32)
    for bi in range(4):
        prog = render(macros[f"A4_DUMP_{bi}"], SUB)
        d = Block(prog, gmem, False, False)
        d.lds = blk.lds
        for w, w0 in zip(d.waves, blk.waves):
            w.a = w0.a
            lane = np.arange(64)
            wm, wn = w.wid >> 1, w.wid & 1
            row = wm * 32 + (lane & 31)
            w.v[121] = (row * 1040 + (wn * 128 + 16 * (lane >> 5)) * 4).astype(np.uint32)
        d.run(sched="fwd")
        cs = blk.lds[:64 * 1040].view(np.float32).reshape(64, 260)
        for lr in range(64):
            got[(lr >> 5) * 128 + bi * 32 + (lr & 31), :] = cs[lr, :256]
    mm, nn = min(256, M - m0), min(256, N - n0)
    err = np.abs(got[:mm, :nn].astype(np.float64) - ref[m0:m0 + mm, n0:n0 + nn]).max()
    if verbose:
        print(f"ta={ta} tb={tb} M={M} N={N} K={K} tile={tile} lazy_ds={lazy_ds} lazy_dma={lazy_dma} sched={sched}: max abs err {err:.3e}")
    return err


SUB_P = dict(tid="v250", pa0="s8", pa1="s9", pb0="s10", pb1="s11", lda="s12", ldb="s13", pc0="s14", pc1="s15", ldc="s16", cbytes="s17",
             niter="s18", lds="s19", bid="s20", grid="s21", q="s22", r="s23", magicg="s24", gsz="s25", magicm="s100", magicl="s101", walk="s102", nmy="s26", mlast="s27", nlast="s28", pz0="s29", pz1="s30", zbytes="s31", scale="s32", hseed="s33", hpp="s34")


def keep_mask(M, N, seed, p16, row0=0):
    """numpy restatement of v2s_keep8 (csrc/v2s_common.h) for a whole [M][N] matrix (N % 8 == 0, (row0 + M) * N < 2^35): True = kept"""
    u32 = lambda x: np.asarray(x, np.uint64) & np.uint64(0xFFFFFFFF)
    chunk = (np.arange(M, dtype=np.uint64)[:, None] + np.uint64(row0)) * np.uint64(N // 8) + np.arange(N // 8, dtype=np.uint64)[None, :]
    x = u32(np.uint64((seed * 0x9E3779B1) & 0xFFFFFFFF) + chunk)
    x = u32(x * np.uint64(0x9E3779B1)); x ^= x >> np.uint64(15); x = u32(x * np.uint64(0x85EBCA77)); x ^= x >> np.uint64(13)
    keep = np.zeros((M, N // 8, 8), bool)
    for i in range(4):
        wv = u32((x >> np.uint64(8 * i)) | (x << np.uint64(32 - 8 * i))) if i else x
        h = u32((wv & np.uint64(0xFFFFFF)) * np.uint64(0x00EBCA77 + 0x2468 * i))
        keep[:, :, 2 * i] = (h & np.uint64(0xFFFF)) >= p16
        keep[:, :, 2 * i + 1] = (h >> np.uint64(16)) >= p16
    return keep.reshape(M, N)


def walk_args(tilesM, tilesN, GM):
    """scalar arguments of the grouped tile walk (mirrors the host code in v2s_gemm_a4.h): gsz, magic(gsz), magic(GM), magic(gm_last), packed word"""
    magic = lambda d: (((1 << 32) + d - 1) // d) if d >= 2 else 0
    GM = max(1, min(GM, tilesM, 255))
    gm_last = tilesM % GM
    glast = tilesM // GM if gm_last else 0xFFFF
    return GM * tilesN, magic(GM * tilesN), magic(GM), magic(gm_last), GM | (gm_last << 8) | (glast << 16)


def check_p(inc, tb, M, N, K, grid, seed=1, lazy_ds=True, lazy_dma=True, sched="random", verbose=False, dact=False, scale=1.0, epi="", p16=6554, hseed=0x1234567, GM=1):
    """the persistent deferred-write-out kernel: `grid` blocks walk the (M / 256) x (N / 256) tiles; returns the max abs error of the bf16
    output against the fp64 product rounded to bf16 inputs"""
    macros = parse_inc(inc)
    rs = np.random.RandomState(seed)
    X = (rs.randn(M, K) * 0.5).astype(np.float32)
    W = (rs.randn(N, K) * 0.5).astype(np.float32)
    to_bf = lambda f: ((f.view(np.uint32) + 0x7FFF + ((f.view(np.uint32) >> 16) & 1)) >> 16).astype(np.uint16)
    Xb, Wb = to_bf(X), to_bf(W)
    ref = bf16_to_f32(Xb).astype(np.float64) @ bf16_to_f32(Wb).astype(np.float64).T
    lda = K
    if tb:
        Bmem, ldb = np.ascontiguousarray(Wb.T), N
    else:
        Bmem, ldb = Wb, K
    ldc = N
    Cmem = np.full(M * ldc, 0x7FC0, np.uint16)            # NaN-filled output
    PA, PB, PC, PZ = 0x10000000, 0x30000000, 0x50000000, 0x70000000
    gmem = [(PA, Xb.view(np.uint8).reshape(-1)), (PB, Bmem.view(np.uint8).reshape(-1)), (PC, Cmem.view(np.uint8))]
    if dact:                                              # z: the forward's post-dropout ReLU output: zeros (incl. -0) and positive values, some negative
        zf = rs.randn(M, ldc).astype(np.float32)
        zf[rs.rand(M, ldc) < 0.4] = 0.0
        Zb = to_bf(zf)
        Zb[rs.rand(M, ldc) < 0.02] = 0x8000              # -0
        gmem.append((PZ, Zb.view(np.uint8).reshape(-1)))
    tilesM, tilesN = (M + 255) // 256, (N + 255) // 256
    ntiles = tilesM * tilesN
    if epi:
        assert not tb and not dact
    prog = render(macros[f"A4P_MAIN_NT_{epi.upper()}" if epi else ("A4P_MAIN_NN_DACT" if dact else ("A4P_MAIN_NN" if tb else "A4P_MAIN_NT"))], SUB_P)
    for bid in range(grid):
        blk = Block(prog, gmem, lazy_ds, lazy_dma)
        for w in blk.waves:
            w.a[:] = np.float32(np.nan).view(np.uint32)   # garbage accumulators at kernel start
            w.v[250] = np.arange(64, dtype=np.uint32) + 64 * w.wid
            w.s[8], w.s[9] = PA & 0xFFFFFFFF, PA >> 32
            w.s[10], w.s[11] = PB & 0xFFFFFFFF, PB >> 32
            w.s[12], w.s[13] = lda * 2, ldb * 2
            w.s[14], w.s[15] = PC & 0xFFFFFFFF, PC >> 32
            w.s[16], w.s[17] = ldc * 2, M * ldc * 2
            w.s[18], w.s[19] = K // 128, 0
            w.s[20], w.s[21] = bid, grid
            w.s[22], w.s[23] = ntiles >> 3, ntiles & 7
            gsz, mg, mm, ml, wk = walk_args(tilesM, tilesN, GM)
            w.s[24], w.s[25], w.s[100], w.s[101], w.s[102] = mg, gsz, mm, ml, wk
            w.s[26] = (ntiles - bid + grid - 1) // grid
            w.s[27], w.s[28] = M - 256, N - 256
            w.s[29], w.s[30], w.s[31] = PZ & 0xFFFFFFFF, PZ >> 32, M * ldc * 2
            w.s[32] = int(np.float32(scale).view(np.uint32))
            w.s[33] = (hseed * 0x9E3779B1) & 0xFFFFFFFF                       # seed term (row0 = 0)
            w.s[34] = (((p16 ^ 0x8000) - 1) & 0xFFFF) * 0x10001
        blk.run(sched=sched, seed=seed + bid)
        assert all(not w.vm and not w.lgkm for w in blk.waves)
    got = bf16_to_f32(Cmem).reshape(M, ldc).astype(np.float64)
    if dact:
        ref = np.where(bf16_to_f32(Zb).reshape(M, ldc)[:, :N] > 0, ref.astype(np.float32) * np.float32(scale), 0.0).astype(np.float64)
    atol = 1e-6
    if epi:                                               # ReLU (+ dropout with the library's counter-based mask, scale 1 / (1 - p))
        ref = np.maximum(ref, 0.0)
        if epi == "reludrop":
            ref = np.where(keep_mask(M, N, hseed, p16), ref.astype(np.float32) * np.float32(scale), 0.0).astype(np.float64)
        atol = 2e-4                                       # a sum within fp32 rounding of zero may land on either side of the ReLU
    refb = bf16_to_f32(to_bf(ref.astype(np.float32))).astype(np.float64)
    err = np.abs(got - refb)
    bad = ~(err <= 0.0079 * np.abs(refb) + atol)          # one bf16 ulp: the fp32 sums differ from fp64 in the last bits
    if verbose:
        print(f"persistent tb={tb} M={M} N={N} K={K} grid={grid} lazy_ds={lazy_ds} lazy_dma={lazy_dma} sched={sched}: "
              f"{int(bad.sum())} wrong of {bad.size}, NaN {int(np.isnan(got).sum())}, max abs err {np.nanmax(err):.3e}")
    return int(bad.sum())


if __name__ == "__main__":
    import os
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ta = "--tn" in sys.argv
    tb = "--nn" in sys.argv or ta
    inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vidchapters_amd", "csrc", "v2s_gemm_a4.inc")
    for a in sys.argv[1:]:
        if a.startswith("--inc="):
            inc = a[6:]
    M, N, K = (int(x) for x in args[:3]) if len(args) >= 3 else (256, 256, 128)
    if "--persistent" in sys.argv:
        grid = int(args[3]) if len(args) > 3 else 2
        nbad = 0
        for lazy_ds, lazy_dma, sched in ((False, False, "fwd"), (True, True, "random"), (True, False, "random"), (False, True, "rev"), (True, True, "fwd")):
            epi = "reludrop" if "--reludrop" in sys.argv else ("relu" if "--relu" in sys.argv else "")
            nbad += check_p(inc, tb, M, N, K, grid, lazy_ds=lazy_ds, lazy_dma=lazy_dma, sched=sched, verbose=True, dact="--dact" in sys.argv,
                            scale=1.0 / 0.9 if ("--dact" in sys.argv or epi == "reludrop") else 1.0, epi=epi)
        print("OK" if nbad == 0 else "FAILED")
        sys.exit(0 if nbad == 0 else 1)
    worst = 0.0
    for lazy_ds, lazy_dma, sched in ((False, False, "fwd"), (True, True, "random"), (True, False, "random"), (False, True, "rev"), (True, True, "fwd")):
        worst = max(worst, check(inc, tb, M, N, K, tile=((M - 1) // 256, (N - 1) // 256), lazy_ds=lazy_ds, lazy_dma=lazy_dma, sched=sched, verbose=True, ta=ta))
    print("OK" if worst < 1e-3 * (K ** 0.5) else "FAILED")
