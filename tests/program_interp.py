"""
TEST INFRASTRUCTURE: a numpy interpreter of the ghn3_op programs (include/ghn3_hip.h semantics).

It lets the CPU test-suite validate the host-side compiler (ghn3_amd/program.py: op order, buffer offsets,
GEMM addressing modes, tile descriptors, the whole backward program) against the oracle without a GPU.  It is
never imported by the package and is not a fallback: ghn3_amd raises without libghn3_hip.so + a HIP device.
"""

import math
import numpy as np
from scipy.special import erf

from ghn3_amd import _lib as L


class Interp:
    def __init__(self, bufs):
        """bufs: list of numpy uint8 arrays (or None), indexed like the device pointer table."""
        self.bufs = bufs

    # ---- memory views -----------------------------------------------------------------------------
    def view(self, ref, dtype, count):
        buf, off = int(ref['buf']), int(ref['off'])
        if buf < 0:
            return None
        b = self.bufs[buf]
        assert b is not None, 'absent buffer %d' % buf
        item = np.dtype(dtype).itemsize
        assert off % min(item, 16) == 0
        out = b[off:off + count * item].view(dtype)
        assert len(out) == count, (buf, off, count, len(b))
        return out

    def fview(self, ref, count):
        return self.view(ref, np.float32, count)

    def tail(self, ref, dtype, before=0):
        """view from the reference to the end of its buffer (`before`: start that many elements in front of it)"""
        buf, off = int(ref['buf']), int(ref['off'])
        if buf < 0:
            return None
        off -= before * np.dtype(dtype).itemsize
        b = self.bufs[buf]
        n = (len(b) - off) // np.dtype(dtype).itemsize
        return b[off:off + n * np.dtype(dtype).itemsize].view(dtype)

    # ---- ops -----------------------------------------------------------------------------------------
    def run(self, ops, problems):
        for o in ops:
            kind = int(o['kind'])
            getattr(self, 'op_' + L.OP_NAMES[kind])(o, problems)

    def op_nop(self, o, problems):
        pass

    def op_join(self, o, problems):
        pass                         # (the interpreter runs side-stream ops in program order)

    def op_detach(self, o, problems):
        pass

    def op_rowset_colsum(self, o, problems):
        n_sets, O, I = (int(v) for v in o['i'][:3])
        out = self.tail(o['r'][0], np.float32)
        X = self.tail(o['r'][1], np.float32)
        sets = self.view(o['r'][2], L.ROWSET_DT, n_sets)
        acc = np.zeros((O, I), np.float64)
        for S in sets:
            off, rows, o_, i_, ld, i0 = (int(S[k]) for k in ('off', 'rows', 'o', 'i', 'ld', 'i0'))
            blk = X[off + np.arange(rows)[:, None] * ld + np.arange(o_ * i_)[None, :]].astype(np.float64).sum(0)
            acc[:o_, i0:i0 + i_] += blk.reshape(o_, i_)
        out[:O * I] += acc.reshape(-1).astype(np.float32)

    def op_relu_fix(self, o, problems):
        rows, cols, ld, K, q, s_ = (int(v) for v in o['i'][:6])
        X = self.tail(o['r'][0], np.float32)
        U = self.tail(o['r'][1], np.float32)
        W = self.tail(o['r'][2], np.float32)
        b = self.tail(o['r'][3], np.float32)
        tau_rel = float(o['f'][0])
        c = np.arange(cols)
        wrow = (c // q) * s_ + c % q if q > 0 else c
        for r in range(rows):
            x = X[r * ld:r * ld + cols]
            if tau_rel > 0:
                for c0 in range(0, cols, 1024):
                    seg = x[c0:c0 + 1024]
                    tau = tau_rel * np.sqrt(float((seg.astype(np.float64) ** 2).mean()))
                    for k in np.nonzero(np.abs(seg) < tau)[0]:
                        wr = wrow[c0 + k]
                        v = float(W[wr * K:(wr + 1) * K].astype(np.float64) @ U[r * K:(r + 1) * K].astype(np.float64))
                        seg[k] = v + (float(b[wr]) if b is not None else 0.0)
            np.maximum(x, 0, out=x)

    def op_sumsq(self, o, problems):
        n = int(o['i'][0])
        x = self.fview(o['r'][1], n).astype(np.float64)
        self.fview(o['r'][0], 1)[0] += np.float32((x * x).sum())

    def op_adamw(self, o, problems):
        n = int(o['i'][0])
        lr, b1, b2, eps, wd, bc1, bc2 = (float(v) for v in o['i'][1:8].view(np.float64))
        p, g, m, v = (self.fview(o['r'][k], n) for k in range(4))
        inv = float(o['f'][1]) if float(o['f'][1]) > 0 else 1.0
        clip = inv
        if int(o['r'][4]['buf']) >= 0:
            if not np.isfinite(self.fview(o['r'][4], 1)[0]):
                return
            if float(o['f'][0]) > 0:
                clip *= min(1.0, float(o['f'][0]) / (float(np.sqrt(self.fview(o['r'][4], 1)[0])) * inv + 1e-6))
        gi = g.astype(np.float64) * clip
        m[:] = b1 * m + (1 - b1) * gi
        v[:] = b2 * v + (1 - b2) * gi * gi
        p[:] = p * (1 - lr * wd) - (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)

    @staticmethod
    def _rowmap(r, g, q, s):
        r = np.asarray(r, dtype=np.int64)
        if g is not None:
            r = g[r].astype(np.int64)
        if q > 0:
            r = (r // q) * s + (r % q)
        return r

    # ---- 16-bit operand copies ------------------------------------------------------------------------
    @staticmethod
    def to16(x, bf16):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if not bf16:
            return x.astype(np.float16).view(np.uint16)
        u = x.view(np.uint32).astype(np.uint64)
        return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) & 0xffff).astype(np.uint16)

    @staticmethod
    def from16(h, bf16):
        h = np.ascontiguousarray(h, dtype=np.uint16)
        if not bf16:
            return h.view(np.float16).astype(np.float32)
        return (h.astype(np.uint32) << 16).view(np.float32)

    @staticmethod
    def pow2_scale(amax):
        """ghn3_pow2_scale: 2^(11 - e) for amax = m 2^e."""
        amax = float(amax)
        if not amax > 0:
            return 1.0
        e = int(np.floor(np.log2(amax)))
        return 1.0 if e < -100 else float(2.0 ** (11 - e))

    @staticmethod
    def frag_index(n, k, K):
        """GHN3_CAST_FRAG: position of element (n, k) of a [n][k] matrix with K columns in fragment-major order."""
        return ((n // 16) * (K // 32) + k // 32) * 512 + (((k % 32) // 8) * 16 + n % 16) * 8 + k % 8

    def op_cast16(self, o, problems):
        n_desc, blocks = int(o['i'][0]), int(o['i'][1])
        src = self.tail(o['r'][0], np.float32)
        dst = self.tail(o['r'][1], np.uint16)
        descs = self.view(o['r'][2], L.CAST_DT, n_desc)
        dbias = self.tail(o['r'][3], np.float32)
        amax = self.tail(o['r'][4], np.float32)
        r64 = lambda v: (v + 63) // 64 * 64
        nb = 0
        for D in descs:
            rows, cols, ld = int(D['rows']), int(D['cols']), int(D['ld_src'])
            assert int(D['block_start']) == nb
            nb += ((rows + 63) // 64) * ((cols + 63) // 64)
            off = int(D['src_off'])
            cc = np.arange(cols)
            if int(D['src_q']) > 0:
                assert int(D['src_q']) % 4 == 0 and int(D['src_s']) % 4 == 0
                cc = (cc // int(D['src_q'])) * int(D['src_s']) + cc % int(D['src_q'])
            fl = int(D['flags'])
            if fl & L.CAST_SRC16:
                # 16-bit source inside the destination buffer, already scaled: re-laid out as is; sums unscaled again
                assert not fl & (L.CAST_STRAIGHT | L.CAST_SPLIT)
                X = self.from16(dst[off + np.arange(rows)[:, None] * ld + cc[None, :]], bool(fl & L.CAST_TRANSPOSED_BF16))
                Xsum = X
                if fl & L.CAST_SCALED and amax is not None:
                    Xsum = (X / np.float32(self.pow2_scale(amax[0]))).astype(np.float32)
            else:
                X = src[off + np.arange(rows)[:, None] * ld + cc[None, :]]
                Xsum = X
                if fl & L.CAST_SCALED and amax is not None:
                    X = (X * np.float32(self.pow2_scale(amax[0]))).astype(np.float32)
            split = bool(fl & L.CAST_SPLIT)
            f16s = split and bool(fl & L.CAST_SPLIT_F16)      # straight pieces: f16 pieces of x * 2^GHN3_X3F16_WSHIFT
            if split:                               # bf16 hi copy + lo = bf16(x - hi) copy `lo_off` elements behind
                Xlo = (X - self.from16(self.to16(X, True), True)).astype(np.float32)
            if f16s:
                Xs = (X * np.float32(2.0 ** L.X3F16_WSHIFT)).astype(np.float32)
                Xs_lo = (Xs - self.from16(self.to16(Xs, False), False)).astype(np.float32)
            if fl & L.CAST_FRAG:
                assert split and rows % 32 == 0 and cols % 32 == 0
                r_, c_ = np.arange(rows)[:, None], np.arange(cols)[None, :]
                if fl & L.CAST_STRAIGHT:
                    ii = int(D['dst_off']) + self.frag_index(r_, c_, cols)
                    dst[ii] = self.to16(Xs, False) if f16s else self.to16(X, True)
                    dst[ii + int(D['lo_off'])] = self.to16(Xs_lo, False) if f16s else self.to16(Xlo, True)
                if fl & L.CAST_TRANSPOSED:
                    ii = int(D['dstT_off']) + self.frag_index(c_, r_, rows)
                    dst[ii] = self.to16(X, True)
                    dst[ii + int(D['lo_off'])] = self.to16(Xlo, True)
                continue
            if fl & L.CAST_STRAIGHT:
                Z = np.zeros((rows, r64(cols)), np.float32)
                Z[:, :cols] = X
                ldd = int(D['ld_dst'])
                ii = int(D['dst_off']) + np.arange(rows)[:, None] * ldd + np.arange(r64(cols))[None, :]
                if f16s:
                    Z[:, :cols] = Xs
                dst[ii] = self.to16(Z, (bool(fl & L.CAST_STRAIGHT_BF16) or split) and not f16s)
                if split:
                    Z[:, :cols] = Xs_lo if f16s else Xlo
                    dst[ii + int(D['lo_off'])] = self.to16(Z, not f16s)
            if fl & L.CAST_TRANSPOSED:
                rw = (rows + 7) // 8 * 8 if fl & L.CAST_TIGHT else r64(rows)
                Z = np.zeros((cols, rw), np.float32)
                Z[:, :rows] = X.T
                ldd = int(D['ld_dstT'])
                ii = int(D['dstT_off']) + np.arange(cols)[:, None] * ldd + np.arange(rw)[None, :]
                dst[ii] = self.to16(Z, bool(fl & L.CAST_TRANSPOSED_BF16) or split)
                if split:
                    Z[:, :rows] = Xlo.T
                    dst[ii + int(D['lo_off'])] = self.to16(Z, True)
            if fl & L.CAST_COLSUM_PARTS:
                for rt in range((rows + 63) // 64):
                    dbias[int(D['part_off']) + rt * cols + np.arange(cols)] = \
                        Xsum[rt * 64:(rt + 1) * 64].astype(np.float64).sum(0).astype(np.float32)
            elif fl & L.CAST_COLSUM:
                c = np.arange(cols)
                q, s_ = int(D['bias_q']), int(D['bias_s'])
                if q > 0:
                    c = (c // q) * s_ + c % q
                np.add.at(dbias, c + int(D['bias_off']), Xsum.astype(np.float64).sum(0).astype(np.float32))
        assert nb == blocks

    def _gemm_op16(self, o, p):
        ct = (int(o['flags']) & 0xff) - 1
        assert ct in (L.CT_F16, L.CT_BF16), 'OP16 problems need an explicit 16-bit compute type on the op'
        bf = ct == L.CT_BF16
        M, N, K = int(p['M']), int(p['N']), int(p['K'])
        lda, ldb = int(p['lda']), int(p['ldb'])
        assert int(p['a_mode']) == L.MODE_ROW and int(p['b_mode']) == L.MODE_ROW and lda % 8 == 0 and ldb % 8 == 0
        assert int(p['A']['off']) % 16 == 0 and int(p['B']['off']) % 16 == 0 and int(p['b_kq']) % 8 == 0
        XA, XB = self.tail(p['A'], np.uint16), self.tail(p['B'], np.uint16)
        ga, gb = self.tail(p['a_gather'], np.int32), self.tail(p['b_gather'], np.int32)
        Kp = (K + 63) // 64 * 64
        k = np.arange(Kp)                     # the kernel reads whole 64-wide k tiles
        ra = self._rowmap(np.arange(M), ga, int(p['a_q']), int(p['a_s']))
        rb = self._rowmap(np.arange(N), gb, int(p['b_q']), int(p['b_s']))
        kq, ks = int(p['b_kq']), int(p['b_ks'])
        kphys = (k // kq) * ks + k % kq if kq > 0 else k
        A = self.from16(XA[ra[:, None] * lda + k[None, :]], bf)
        Bm = self.from16(XB[rb[None, :] * ldb + kphys[:, None]], bf)
        assert np.all(A[:, K:] == 0), 'A must hold zeros in the K padding'
        assert np.isfinite(Bm).all()
        return A, Bm

    # (tile code, ln_kind) -> reduction lengths the staged kernels are instantiated for (gemm_x3d.hip g_cfg)
    X3S_K = {44: {0: (64, 128, 192, 256, 384), 1: (64, 128, 192, 256, 384), 2: (64, 128, 192, 256, 384)},
             45: {0: (64, 128, 192, 256, 384, 512, 768, 1024, 1152, 1536), 1: (64, 128, 192, 256, 384),
                  2: (64, 128, 192, 256, 384)}}

    def _gemm_x3(self, p, tile=40):
        """GHN3_GEMM_X3: fp32 A split on the fly into bf16 hi + lo, B from the bf16 hi / lo copies; the three products
        hi.hi + hi.lo + lo.hi (fp32 accumulate on the device, fp64 here).  Tile codes 44 / 45: fragment-major copies,
        optional LayerNorm row prologue of A."""
        M, N, K = int(p['M']), int(p['N']), int(p['K'])
        lda, ldb, sl = int(p['lda']), int(p['ldb']), int(p['x3_slice'])
        staged = tile in (44, 45)
        assert int(p['a_mode']) == L.MODE_ROW and int(p['b_mode']) == L.MODE_ROW and ldb % 8 == 0 and N % 4 == 0
        if staged:
            assert K in self.X3S_K[tile][int(p['ln_kind'])] and N % 16 == 0 and int(p['ksplit']) <= 1, (tile, K, N)
        else:
            assert sl > 0 and sl % 64 == 0 and K % sl == 0 and sl <= 384 and int(p['ksplit']) <= 1
            assert not int(p['ln_kind'])
        assert int(p['b_gather']['buf']) < 0
        assert int(p['B']['off']) % 16 == 0 and int(p['B2']['off']) % 16 == 0 and int(p['B2']['buf']) >= 0
        XA = self.tail(p['A'], np.float32)
        Bh, Bl = self.tail(p['B'], np.uint16), self.tail(p['B2'], np.uint16)
        ga = self.tail(p['a_gather'], np.int32)
        ra = ga[:M].astype(np.int64) if ga is not None else np.arange(M)
        A = XA[ra[:, None] * lda + np.arange(K)[None, :]]
        if int(p['ln_kind']):
            A = self._ln_prologue(p, A, M, K, lda)
        bf = not (int(p['flags']) & L.GEMM_X3F16)          # GHN3_GEMM_X3F16: f16 pieces (B already scaled, alpha undoes it)
        assert bf or staged
        ah = self.from16(self.to16(A, bf), bf)
        al = self.from16(self.to16((A - ah).astype(np.float32), bf), bf)
        if staged:
            ib = self.frag_index(np.arange(N)[None, :], np.arange(K)[:, None], K)
        else:
            ib = np.arange(N)[None, :] * ldb + np.arange(K)[:, None]
        bh, bl = self.from16(Bh[ib], bf).astype(np.float64), self.from16(Bl[ib], bf).astype(np.float64)
        ah, al = ah.astype(np.float64), al.astype(np.float64)
        self._gemm_finish(p, None, None, prod=ah @ bh + (ah @ bl + al @ bh))

    def op_gemm(self, o, problems):
        first, cnt = int(o['i'][0]), int(o['i'][1])
        for p in problems[first:first + cnt]:
            M, N, K = int(p['M']), int(p['N']), int(p['K'])
            if M <= 0 or N <= 0:
                continue
            if int(p['flags']) & L.GEMM_X3:
                self._gemm_x3(p, int(o['i'][2]))
                continue
            if int(p['flags']) & L.GEMM_OP16:
                A, Bm = self._gemm_op16(o, p)
                col_ext = None
                if int(p['lim']['buf']) >= 0:
                    lim = self.tail(p['lim'], np.int32)[:(M + 127) // 128]
                    ext = np.repeat(lim, 128)[:M]
                    assert np.all(np.diff(lim) <= 0), 'rows must be sorted by decreasing extent'
                    if int(p['lim_kind']) == 2:
                        # reduction extents: A must already hold zeros beyond each row block's extent
                        kk = np.arange(A.shape[1])[None, :]
                        assert np.all(A[kk >= ext[:, None] * np.ones_like(kk)] == 0)
                    else:
                        col_ext = ext           # columns >= ext[row] are don't-care: the interpreter leaves them
                self._gemm_finish(p, A, Bm, col_ext)
                continue
            lda, ldb, ldc = int(p['lda']), int(p['ldb']), int(p['ldc'])
            XA, XB, Y = self.tail(p['A'], np.float32), self.tail(p['B'], np.float32), self.tail(p['C'], np.float32)
            ga, gb, gc = (self.tail(p[n], np.int32) for n in ('a_gather', 'b_gather', 'c_gather'))
            assert lda % 4 == 0 and ldb % 4 == 0 and int(p['A']['off']) % 16 == 0 and int(p['B']['off']) % 16 == 0
            if int(p['a_mode']) == L.MODE_ROW:
                ra = self._rowmap(np.arange(M), ga, int(p['a_q']), int(p['a_s']))
                A = XA[(ra[:, None] * lda + np.arange(K)[None, :])]
                if int(p['ln_kind']):
                    A = self._ln_prologue(p, A, M, K, lda)
            else:
                ra = self._rowmap(np.arange(K), ga, int(p['a_q']), int(p['a_s']))
                A = XA[(ra[None, :] * lda + np.arange(M)[:, None])]
            if int(p['b_mode']) == L.MODE_ROW:
                rb = self._rowmap(np.arange(N), gb, int(p['b_q']), int(p['b_s']))
                Bm = XB[(rb[None, :] * ldb + np.arange(K)[:, None])]
            else:
                rb = self._rowmap(np.arange(K), gb, int(p['b_q']), int(p['b_s']))
                Bm = XB[(rb[:, None] * ldb + np.arange(N)[None, :])]
            self._gemm_finish(p, A, Bm)

    def _ln_prologue(self, p, A, M, K, lda):
        """ghn3_gemm_problem::ln_kind: LayerNorm forward (1) / backward (2) applied to the rows of A."""
        kind = int(p['ln_kind'])
        assert int(p['a_gather']['buf']) < 0 and int(p['a_q']) == 0 and K % 4 == 0 and K <= 4096
        P = [self.tail(p['ln_p'][e], np.float32) for e in range(6)]
        idx = np.arange(M)[:, None] * lda + np.arange(K)[None, :]
        A64 = A.astype(np.float64)
        g = P[0][:K].astype(np.float64)
        if kind == 1:
            mu = A64.mean(1, keepdims=True)
            rs = 1.0 / np.sqrt(((A64 - mu) ** 2).mean(1, keepdims=True) + float(p['ln_eps']))
            out = (A64 - mu) * rs * g + P[1][:K].astype(np.float64)
            if P[2] is not None:
                P[2][:M] = mu[:, 0]
            if P[3] is not None:
                P[3][:M] = rs[:, 0]
            if P[4] is not None:
                P[4][idx] = out.astype(np.float32)
        else:
            assert kind == 2
            x = P[1][idx].astype(np.float64)
            mu = P[2][:M].astype(np.float64)[:, None]
            rs = P[3][:M].astype(np.float64)[:, None]
            xh = (x - mu) * rs
            dg = A64 * g
            s1 = dg.mean(1, keepdims=True)
            s2 = (dg * xh).mean(1, keepdims=True)
            out = rs * (dg - s1 - xh * s2)
            if P[4] is not None:
                out = out + P[4][idx].astype(np.float64)
            if P[5] is not None:
                P[5][idx] = out.astype(np.float32)
        return out.astype(np.float32)

    def _gemm_finish(self, p, A, Bm, col_ext=None, prod=None):
        if True:
            M, N = int(p['M']), int(p['N'])
            ldc = int(p['ldc'])
            Y = self.tail(p['C'], np.float32)
            gc = self.tail(p['c_gather'], np.int32)
            alpha = float(p['alpha'])
            if 'alpha_amax' in p.dtype.names and int(p['alpha_amax']['buf']) >= 0:
                alpha /= self.pow2_scale(self.tail(p['alpha_amax'], np.float32)[0])
            v = (prod if prod is not None else (A.astype(np.float64) @ Bm.astype(np.float64))) * alpha
            rc = self._rowmap(np.arange(M), gc, int(p['c_q']), int(p['c_s']))
            ci = rc[:, None] * ldc + np.arange(N)[None, :]
            if int(p['flags']) & L.GEMM_BIASGRAD:
                assert int(p['a_mode']) == L.MODE_COL
                db = self.tail(p['bias'], np.float32)
                np.add.at(db, rc * (int(p['bias_stride']) or 1), A.astype(np.float64).sum(1).astype(np.float32))
            elif int(p['bias']['buf']) >= 0:
                bias = self.tail(p['bias'], np.float32)
                n = np.arange(N)
                bi = n
                if int(p['bias_q']) > 0:
                    bi = (n // int(p['bias_q'])) * int(p['bias_s']) + n % int(p['bias_q'])
                stride = int(p['bias_stride']) or 1
                v = v + bias[bi * stride].astype(np.float64)[None, :]
            if int(p['flags']) & getattr(L, 'GEMM_SUMSQ', 16):
                # GHN3_GEMM_SUMSQ: the slots of the table add up to the sum of the squares stored (the kernel's slot of a
                # value is an implementation detail: here every problem adds into slot 0)
                self.tail(p['aux_out'], np.float32)[0] += np.float32((v.astype(np.float32).astype(np.float64) ** 2).sum())
            elif int(p['aux_out']['buf']) >= 0:
                self.tail(p['aux_out'], np.float32)[ci] = v.astype(np.float32)
            act = int(p['act'])
            if act == L.ACT_RELU:
                v = np.maximum(v, 0)
            elif act == L.ACT_GELU:
                v = 0.5 * v * (1 + erf(v / math.sqrt(2)))
            dact = int(p['dact'])
            if dact != L.DACT_NONE:
                z = self.tail(p['aux_in'], np.float32)[ci].astype(np.float64)
                if dact == L.DACT_RELU:
                    v = np.where(z > 0, v, 0.0)
                else:
                    v = v * (0.5 * (1 + erf(z / math.sqrt(2))) + z * np.exp(-0.5 * z * z) / math.sqrt(2 * math.pi))
            if int(p['residual']['buf']) >= 0:
                v = v + self.tail(p['residual'], np.float32)[ci]
            if int(p['flags']) & L.GEMM_ACCUM or int(p['ksplit']) > 1:
                v = v + Y[ci]
            if col_ext is not None:
                keep = np.arange(N)[None, :] < col_ext[:, None]
                Y[ci[keep]] = v[keep].astype(np.float32)
            else:
                Y[ci] = v.astype(np.float32)

    def op_graph_prologue(self, o, problems):
        B, N, V = (int(v) for v in o['i'][:3])
        A = self.view(o['r'][0], np.int64, B * N * N).reshape(B, N, N)
        one = (A == 1)
        self.view(o['r'][1], np.int32, B * N)[:] = np.minimum(one.sum(1), 100).reshape(-1)
        self.view(o['r'][2], np.int32, B * N)[:] = np.minimum(one.sum(2), 100).reshape(-1)
        self.view(o['r'][3], np.int32, B * N)[:] = np.clip(A[:, 0, :], 0, 1000).reshape(-1)
        f = np.clip(A, 0, V - 1)
        self.view(o['r'][4], np.int32, B * N * N)[:] = (f * V + f.transpose(0, 2, 1)).reshape(-1)

    def _embed_common(self, o):
        B, N, C = (int(v) for v in o['i'][:3])
        n_nodes = self.view(o['r'][3], np.int32, B)
        total = int(n_nodes.sum())
        types = self.view(o['r'][1], np.int32, total)
        shp = self.view(o['r'][2], np.int32, 4 * total).reshape(total, 4)
        noff = self.view(o['r'][4], np.int32, B)
        deg_in = self.view(o['r'][11], np.int32, B * N)
        deg_out = self.view(o['r'][12], np.int32, B * N)
        dist0 = self.view(o['r'][13], np.int32, B * N)
        return B, N, C, n_nodes, types, shp, noff, deg_in, deg_out, dist0

    def op_embed_nodes(self, o, problems):
        B, N, C, n_nodes, types, shp, noff, deg_in, deg_out, dist0 = self._embed_common(o)
        cq = C // 4
        x = self.fview(o['r'][0], B * N * C).reshape(B * N, C)
        T = [self.tail(o['r'][k], np.float32) for k in range(5, 11)]
        rs = lambda v, w: v[:len(v) // w * w].reshape(-1, w)      # (table tails run to the end of the flat buffer)
        Et, Ech, Esp = rs(T[0], C), rs(T[1], cq), rs(T[2], cq)
        Ein, Eout, Ed = rs(T[3], C), rs(T[4], C), rs(T[5], C)
        for b in range(B):
            for i in range(N):
                row = b * N + i
                if i >= n_nodes[b]:
                    x[row] = 0
                    continue
                s = noff[b] + i
                v = Et[types[s]] + np.concatenate([Ech[shp[s, 0]], Ech[shp[s, 1]], Esp[shp[s, 2]], Esp[shp[s, 3]]])
                v = v + Ein[deg_in[row]]
                v = v + Eout[deg_out[row]]
                v = v + Ed[dist0[row]]
                x[row] = v

    def op_embed_bwd(self, o, problems):
        B, N, C, n_nodes, types, shp, noff, deg_in, deg_out, dist0 = self._embed_common(o)
        cq = C // 4
        dx = self.fview(o['r'][0], B * N * C).reshape(B * N, C)
        T = [self.tail(o['r'][k], np.float32) for k in range(5, 11)]

        def tab(t, w):
            n = len(t) // w
            return t[:n * w].reshape(n, w)
        Et, Ech, Esp, Ein, Eout, Ed = tab(T[0], C), tab(T[1], cq), tab(T[2], cq), tab(T[3], C), tab(T[4], C), \
            tab(T[5], C)
        for b in range(B):
            for i in range(int(n_nodes[b])):
                row = b * N + i
                s = noff[b] + i
                g = dx[row]
                Et[types[s]] += g
                Ech[shp[s, 0]] += g[:cq]
                Ech[shp[s, 1]] += g[cq:2 * cq]
                Esp[shp[s, 2]] += g[2 * cq:3 * cq]
                Esp[shp[s, 3]] += g[3 * cq:]
                Ein[deg_in[row]] += g
                Eout[deg_out[row]] += g
                Ed[dist0[row]] += g

    def op_edge_hidden(self, o, problems):
        V, C = int(o['i'][0]), int(o['i'][1])
        Pfw = self.fview(o['r'][1], V * C).reshape(V, C)
        Pbw = self.fview(o['r'][2], V * C).reshape(V, C)
        hid = self.fview(o['r'][0], V * V * C).reshape(V, V, C)
        hid[:] = np.maximum(Pfw[:, None, :] + Pbw[None, :, :], 0)

    def op_edge_hidden_bwd(self, o, problems):
        V, C = int(o['i'][0]), int(o['i'][1])
        dhid = self.fview(o['r'][2], V * V * C).reshape(V, V, C)
        hid = self.fview(o['r'][3], V * V * C).reshape(V, V, C)
        dhid[:] = np.where(hid > 0, dhid, 0)
        self.fview(o['r'][0], V * C).reshape(V, C)[:] = dhid.sum(1)
        self.fview(o['r'][1], V * C).reshape(V, C)[:] = dhid.sum(0)

    def op_bias_gather(self, o, problems):
        B, N, H = (int(v) for v in o['i'][:3])
        ldT = (H + 3) // 4 * 4
        pair = self.view(o['r'][2], np.int32, B * N * N).reshape(B, N, N)
        T = self.tail(o['r'][1], np.float32)
        bias = self.fview(o['r'][0], B * H * N * N).reshape(B, H, N, N)
        for h in range(H):
            bias[:, h] = T[pair.astype(np.int64) * ldT + h]

    def op_bias_hist(self, o, problems):
        B, N, H, V = (int(v) for v in o['i'][:4])
        ldT = (H + 3) // 4 * 4
        pair = self.view(o['r'][2], np.int32, B * N * N).reshape(-1).astype(np.int64)
        dB = self.fview(o['r'][1], B * H * N * N).reshape(B, H, N * N)
        dT = self.fview(o['r'][0], V * V * ldT)
        if int(o['i'][4]):                               # the scale comes from the slot GHN3_OP_ATTN_BWD (r5) filled
            am = self.tail(o['r'][3], np.uint8)[8 * V * V * H:8 * V * V * H + 4].view(np.float32)[0]
            assert abs(float(am) - float(np.abs(dB).max())) <= 1e-6 * float(np.abs(dB).max()) + 1e-30, 'stale max |dBias|'
        for h in range(H):
            np.add.at(dT, pair * ldT + h, dB[:, h, :].reshape(-1))

    def op_layernorm_fwd(self, o, problems):
        rows, C = int(o['i'][0]), int(o['i'][1])
        xv = self.fview(o['r'][1], rows * C).reshape(rows, C)
        if int(o['r'][6]['buf']) >= 0:                  # further K-slice planes of the producing GEMM: summed in place
            n_add, stride = max(1, int(o['i'][2])), int(o['i'][3])
            for q in range(n_add):
                pl = self.tail(o['r'][6], np.float32)[q * stride:q * stride + rows * C].reshape(rows, C)
                xv[:] = xv + pl
        x = xv.astype(np.float64)
        g, b = self.fview(o['r'][2], C), self.fview(o['r'][3], C)
        mu = x.mean(1)
        var = x.var(1)
        rs = 1.0 / np.sqrt(var + float(o['f'][0]))
        self.fview(o['r'][0], rows * C).reshape(rows, C)[:] = ((x - mu[:, None]) * rs[:, None] * g + b)
        if int(o['r'][4]['buf']) >= 0:
            self.fview(o['r'][4], rows)[:] = mu
            self.fview(o['r'][5], rows)[:] = rs

    def op_layernorm_bwd(self, o, problems):
        rows, C = int(o['i'][0]), int(o['i'][1])
        dyv = self.fview(o['r'][1], rows * C).reshape(rows, C)
        if int(o['r'][7]['buf']) >= 0:                  # further K-slice planes of the producing dgrad: summed in place
            n_add, stride = max(1, int(o['i'][2])), int(o['i'][3])
            for q in range(n_add):
                pl = self.tail(o['r'][7], np.float32)[q * stride:q * stride + rows * C].reshape(rows, C)
                dyv[:] = dyv + pl
        dy = dyv.astype(np.float64)
        x = self.fview(o['r'][2], rows * C).reshape(rows, C).astype(np.float64)
        g = self.fview(o['r'][3], C).astype(np.float64)
        mu, rs = self.fview(o['r'][4], rows).astype(np.float64), self.fview(o['r'][5], rows).astype(np.float64)
        xh = (x - mu[:, None]) * rs[:, None]
        dg = dy * g
        dx = rs[:, None] * (dg - dg.mean(1, keepdims=True) - xh * (dg * xh).mean(1, keepdims=True))
        if int(o['r'][6]['buf']) >= 0:
            dx = dx + self.fview(o['r'][6], rows * C).reshape(rows, C)
        self.fview(o['r'][0], rows * C).reshape(rows, C)[:] = dx

    def op_ln_param_grad(self, o, problems):
        rows, C = int(o['i'][0]), int(o['i'][1])
        dy = self.fview(o['r'][2], rows * C).reshape(rows, C).astype(np.float64)
        x = self.fview(o['r'][3], rows * C).reshape(rows, C).astype(np.float64)
        mu, rs = self.fview(o['r'][4], rows), self.fview(o['r'][5], rows)
        xh = (x - mu[:, None]) * rs[:, None]
        dgm, dbt = self.fview(o['r'][0], C), self.fview(o['r'][1], C)
        if not int(o['i'][2]):
            dgm[:] = 0
            dbt[:] = 0
        dgm += (dy * xh).sum(0)
        dbt += dy.sum(0)

    def _attn_common(self, o, nn_ref):
        B, N, C, H = (int(v) for v in o['i'][:4])
        d = C // H
        n_nodes = self.view(nn_ref, np.int32, B)
        valid = np.arange(N)[None, :] < n_nodes[:, None]
        mask = valid[:, :, None] & valid[:, None, :]
        return B, N, C, H, d, mask

    def op_attn_fwd(self, o, problems):
        B, N, C, H, d, mask = self._attn_common(o, o['r'][4])
        qkv = self.fview(o['r'][1], B * N * 3 * C).reshape(B, N, 3, H, d).astype(np.float64)
        q, k, v = (qkv[:, :, j].transpose(0, 2, 1, 3) for j in range(3))
        s = (q @ k.transpose(0, 1, 3, 2)) * d ** -0.5
        if int(o['r'][2]['buf']) >= 0:
            s = s + self.fview(o['r'][2], B * H * N * N).reshape(B, H, N, N)
        s = np.where(mask[:, None], s, -32768.0)
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p = p / p.sum(-1, keepdims=True)
        if int(o['r'][3]['buf']) >= 0:
            self.fview(o['r'][3], B * H * N * N).reshape(B, H, N, N)[:] = p
        out = (p @ v).transpose(0, 2, 1, 3).reshape(B * N, C)
        self.fview(o['r'][0], B * N * C).reshape(B * N, C)[:] = out

    def op_attn_bwd(self, o, problems):
        B, N, C, H, d, mask = self._attn_common(o, o['r'][7])
        scale = d ** -0.5
        qkv = self.fview(o['r'][2], B * N * 3 * C).reshape(B, N, 3, H, d).astype(np.float64)
        q, k, v = (qkv[:, :, j].transpose(0, 2, 1, 3) for j in range(3))
        P = self.fview(o['r'][3], B * H * N * N).reshape(B, H, N, N).astype(np.float64)
        dO = self.fview(o['r'][1], B * N * C).reshape(B, N, H, d).transpose(0, 2, 1, 3).astype(np.float64)
        dV = P.transpose(0, 1, 3, 2) @ dO
        dP = dO @ v.transpose(0, 1, 3, 2)
        # delta = rowsum(P * dP) = rowsum(dO * O): the kernel uses the saved attention output (r4)
        O = self.fview(o['r'][4], B * N * C).reshape(B, N, H, d).transpose(0, 2, 1, 3).astype(np.float64)
        dS = P * (dP - (dO * O).sum(-1, keepdims=True))
        dS = np.where(mask[:, None], dS, 0.0)
        dQ = (dS @ k) * scale
        dK = (dS.transpose(0, 1, 3, 2) @ q) * scale
        if int(o['r'][6]['buf']) >= 0:
            db = self.fview(o['r'][6], B * H * N * N).reshape(B, H, N, N)
            db[:] += dS.astype(np.float32)
            if int(o['r'][5]['buf']) >= 0:               # r5: running max of |dBias| as written by this launch
                am = self.fview(o['r'][5], 1)
                am[0] = max(float(am[0]), float(np.abs(db).max()))
        else:
            assert int(o['r'][5]['buf']) < 0
        out = self.fview(o['r'][0], B * N * 3 * C).reshape(B, N, 3, H, d)
        out[:, :, 0] = dQ.transpose(0, 2, 1, 3)
        out[:, :, 1] = dK.transpose(0, 2, 1, 3)
        out[:, :, 2] = dV.transpose(0, 2, 1, 3)

    @staticmethod
    def _norm(v, mode, scale):
        if mode == 0:
            return v * scale
        if mode == 1:
            return 2.0 / (1.0 + np.exp(-0.5 * v))
        return np.tanh(0.2 * v)

    @staticmethod
    def _norm_grad(v, mode, scale):
        if mode == 0:
            return np.full_like(v, scale)
        if mode == 1:
            sg = 1.0 / (1.0 + np.exp(-0.5 * v))
            return sg * (1 - sg)
        th = np.tanh(0.2 * v)
        return 0.2 * (1 - th * th)

    def _descs(self, o):
        n = int(o['i'][0])
        return self.view(o['r'][7], np.uint8, n * L.TILE_DT.itemsize).view(L.TILE_DT)

    def op_tile_fwd(self, o, problems):
        flat = self.tail(o['r'][0], np.float32)
        srcs = [self.tail(o['r'][1 + k], np.float32) for k in range(6)]
        parts = self.tail(o['r'][8], np.float32)
        bparts = self.tail(o['r'][9], np.float32)
        desc_of_block = None
        if parts is not None:
            nblk = int(o['i'][1])
            tab = self.tail(o['r'][7], np.uint8)[int(o['i'][2]):int(o['i'][2]) + 16 * nblk].view(np.int64).reshape(nblk, 2)
            desc_of_block = np.where(tab[:, 0] < 0, ~tab[:, 0], tab[:, 0])
        kd = 0
        for D in self._descs(o):
            T, E, S = D['T'].astype(np.int64), D['E'].astype(np.int64), D['S'].astype(np.int64)
            idx = [np.arange(T[k]) % E[k] for k in range(4)]
            so = (idx[0][:, None, None, None] * S[0] + idx[1][None, :, None, None] * S[1] +
                  idx[2][None, None, :, None] * S[2] + idx[3][None, None, None, :] * S[3]) + int(D['src_off'])
            v = srcs[int(D['src_buf'])][so.reshape(-1)].astype(np.float64)
            n = int(np.prod(T))
            val = self._norm(v, int(D['mode']), float(D['scale']))
            flat[int(D['dst_off']):int(D['dst_off']) + n] = val
            if parts is not None:                       # r8: per-work-block sums of squares (here: all in the first block)
                blk = np.nonzero(desc_of_block == kd)[0]
                parts[blk] = 0.0
                parts[blk[0]] = np.float32((val.astype(np.float32).astype(np.float64) ** 2).sum())
                if bparts is not None:                  # r9: max |value| * replicas * |scale| (mode 0, source 0)
                    bparts[blk] = 0.0
                    if int(D['src_buf']) == 0 and int(D['mode']) == 0 and val.size:
                        reps = np.prod((T + E - 1) // E)
                        bparts[blk[0]] = np.float32(np.abs(val.astype(np.float32)).max()) * np.float32(abs(float(D['scale']))) * \
                            np.float32(reps)
            kd += 1

    def op_tile_bwd(self, o, problems):
        g = self.tail(o['r'][0], np.float32)
        srcs = [self.tail(o['r'][1 + k], np.float32) for k in range(6)]
        dsrcs = [self.tail(o['r'][8 + k], np.float32) for k in range(5)] + [None]
        amax = self.tail(o['r'][13], np.float32)            # r13: running max |x| written to source-grad buffer 0
        # fused predicted-parameter-norm loss: r14 = per-tensor norms, r15 = predicted values, r6 = device float g,
        # i4 = byte offset (from r7) of the int32 descriptor -> tensor table
        norms = self.tail(o['r'][14], np.float32)
        assert g is not None or norms is not None
        if norms is not None:
            outv, gs = self.tail(o['r'][15], np.float32), float(self.tail(o['r'][6], np.float32)[0])
            dseg = self.tail(o['r'][7], np.uint8)[int(o['i'][4]):int(o['i'][4]) + 4 * int(o['i'][0])].view(np.int32)
        # direct 16-bit tiles: i5 = byte offset (from r7) of {h, rel0, ld32 | ld16 << 32} per descriptor, i6 = bf16
        h16 = None
        if int(o['i'][5]) > 0:
            assert g is None and norms is not None and amax is not None
            h16 = self.tail(o['r'][7], np.uint8)[int(o['i'][5]):int(o['i'][5]) + 24 * int(o['i'][0])].view(np.int64).reshape(-1, 3)
            hdst = self.tail(o['r'][8], np.uint16)
            bound = np.float32(self.tail(o['r'][14], np.float32, before=1)[0]) * np.float32(abs(gs))
            amax[0] = bound
            hsc = np.float32(self.pow2_scale(bound))
        for kd, D in enumerate(self._descs(o)):
            T, E, S, R = (D[k].astype(np.int64) for k in ('T', 'E', 'S', 'R'))
            n = int(np.prod(T))
            gt = np.zeros(tuple(T), dtype=np.float64)
            if g is not None:
                gt = gt + g[int(D['dst_off']):int(D['dst_off']) + n].astype(np.float64).reshape(T)
            if norms is not None:
                nrm = float(norms[int(dseg[kd])])
                if nrm > 0:
                    gt = gt + (gs / nrm) * outv[int(D['dst_off']):int(D['dst_off']) + n].astype(np.float64).reshape(T)
            acc = np.zeros(tuple(R), dtype=np.float64)
            sub = np.zeros(tuple(E), dtype=np.float64)
            idx = [np.arange(T[k]) % E[k] for k in range(4)]
            np.add.at(sub, (idx[0][:, None, None, None], idx[1][None, :, None, None], idx[2][None, None, :, None],
                            idx[3][None, None, None, :]), gt)
            acc[:E[0], :E[1], :E[2], :E[3]] = sub
            ar = [np.arange(R[k]) for k in range(4)]
            so = (ar[0][:, None, None, None] * S[0] + ar[1][None, :, None, None] * S[1] +
                  ar[2][None, None, :, None] * S[2] + ar[3][None, None, None, :] * S[3]) + int(D['src_off'])
            sv = srcs[int(D['src_buf'])][so.reshape(-1)].astype(np.float64).reshape(tuple(R))
            acc = acc * self._norm_grad(sv, int(D['mode']), float(D['scale']))
            inside = np.zeros(tuple(R), dtype=bool)
            inside[:E[0], :E[1], :E[2], :E[3]] = True
            acc = np.where(inside, acc, 0.0)
            if h16 is not None and int(D['src_buf']) == 0 and int(h16[kd, 0]) != np.iinfo(np.int64).min:
                rel = int(h16[kd, 1]) + (so.reshape(-1) - int(D['src_off']))
                ld32, ld16 = int(h16[kd, 2]) & 0xffffffff, int(h16[kd, 2]) >> 32
                assert acc.size == 0 or float(np.abs(acc).max()) <= float(bound) * (1 + 1e-5), 'bound violated'
                hdst[int(h16[kd, 0]) + (rel // ld32) * ld16 + rel % ld32] = \
                    self.to16(acc.reshape(-1).astype(np.float32) * hsc, bool(int(o['i'][6])))
                continue
            dsrcs[int(D['src_buf'])][so.reshape(-1)] = acc.reshape(-1).astype(np.float32)
            if h16 is None and amax is not None and int(D['src_buf']) == 0 and acc.size:
                amax[0] = max(float(amax[0]), float(np.abs(acc.astype(np.float32)).max()))

    def op_ln_param_grad_batch(self, o, problems):
        n, rows, C = int(o['i'][0]), int(o['i'][1]), int(o['i'][2])
        gb, ab = self.tail(o['r'][0], np.float32), self.tail(o['r'][1], np.float32)
        tab = self.view(o['r'][2], np.int64, 6 * n).reshape(n, 6)
        for dg, db, dy, x, mean, rstd in tab:
            dyv = ab[dy:dy + rows * C].reshape(rows, C).astype(np.float64)
            xv = ab[x:x + rows * C].reshape(rows, C).astype(np.float64)
            mu, rs = ab[mean:mean + rows].astype(np.float64)[:, None], ab[rstd:rstd + rows].astype(np.float64)[:, None]
            gb[dg:dg + C] += (dyv * (xv - mu) * rs).sum(0).astype(np.float32)
            gb[db:db + C] += dyv.sum(0).astype(np.float32)

    def op_param_norm_fin(self, o, problems):
        n = int(o['i'][0])
        loss, norms = self.tail(o['r'][0], np.float32), self.tail(o['r'][1], np.float32)
        parts, first = self.tail(o['r'][2], np.float32), self.tail(o['r'][3], np.int32)
        bparts = self.tail(o['r'][4], np.float32)
        for t in range(n):
            norms[t] = np.float32(np.sqrt(parts[int(first[t]):int(first[t + 1])].astype(np.float64).sum()))
        loss[0] = np.float32(norms[:n].astype(np.float64).sum())
        if bparts is not None:                       # r6: max_t (max of t's bound slots) / norms[t]
            ratio, bound = self.tail(o['r'][5], np.float32), self.tail(o['r'][6], np.float32)
            b = np.float32(0)
            for t in range(n):
                bp = bparts[int(first[t]):int(first[t + 1])]
                r = np.float32(bp.max() / norms[t]) if (bp.size and norms[t] > 0) else np.float32(0)
                ratio[t] = r
                b = max(b, r)
            bound[0] = b

    def op_param_norm_fwd(self, o, problems):
        n = int(o['i'][0])
        flat = self.tail(o['r'][1], np.float32)
        seg = self.view(o['r'][2], np.int64, 2 * n).reshape(n, 2)
        assert int(o['i'][1]) >= seg[-1, 1] and np.all(seg[1:, 0] >= seg[:-1, 1]), 'sorted disjoint segments + extent'
        norms = self.fview(o['r'][3], n)
        for s in range(n):
            norms[s] = np.sqrt((flat[seg[s, 0]:seg[s, 1]].astype(np.float64) ** 2).sum())
        self.fview(o['r'][0], 1)[0] += norms.astype(np.float64).sum()

    def op_param_norm_bwd(self, o, problems):
        n = int(o['i'][0])
        flat = self.tail(o['r'][1], np.float32)
        dflat = self.tail(o['r'][0], np.float32)
        seg = self.view(o['r'][2], np.int64, 2 * n).reshape(n, 2)
        assert int(o['i'][1]) >= seg[-1, 1] and np.all(seg[1:, 0] >= seg[:-1, 1]), 'sorted disjoint segments + extent'
        norms = self.fview(o['r'][3], n)
        for s in range(n):
            k = float(o['f'][0]) / norms[s] if norms[s] > 0 else 0.0
            dflat[seg[s, 0]:seg[s, 1]] = flat[seg[s, 0]:seg[s, 1]] * k

    def op_colsum(self, o, problems):
        M, N, ld, q, s, stride, accum = (int(v) for v in o['i'][:7])
        assert accum == 1
        X = self.tail(o['r'][1], np.float32)
        g = self.tail(o['r'][2], np.int32)
        rows = np.arange(M) if g is None else g[:M].astype(np.int64)
        sums = X[rows[:, None] * ld + np.arange(N)[None, :]].astype(np.float64).sum(0)
        n = np.arange(N)
        oi = ((n // q) * s + n % q) if q > 0 else n
        out = self.tail(o['r'][0], np.float32)
        np.add.at(out, oi * (stride or 1), sums.astype(np.float32))

    def op_rowseg_sum(self, o, problems):
        rows, C, ldx, ldo, accum = (int(v) for v in o['i'][:5])
        seg = self.view(o['r'][2], np.int32, rows + 1)
        idx = self.tail(o['r'][3], np.int32)
        X = self.tail(o['r'][1], np.float32)
        out = self.tail(o['r'][0], np.float32)
        for r in range(rows):
            acc = out[r * ldo:r * ldo + C].astype(np.float64) if accum else np.zeros(C)
            for t in range(seg[r], seg[r + 1]):
                acc = acc + X[int(idx[t]) * ldx:int(idx[t]) * ldx + C]
            out[r * ldo:r * ldo + C] = acc

    def op_memset0(self, o, problems):
        n = int(o['i'][0])
        b = self.bufs[int(o['r'][0]['buf'])]
        off = int(o['r'][0]['off'])
        if n > 0:                                            # (unpatched placeholders carry -1: skipped, runtime.hip)
            b[off:off + n] = 0

    def op_dact(self, o, problems):
        M, N, ld, dact = (int(v) for v in o['i'][:4])
        X = self.tail(o['r'][0], np.float32)
        aux = self.tail(o['r'][1], np.float32)
        ii = np.arange(M)[:, None] * ld + np.arange(N)[None, :]
        z = aux[ii].astype(np.float64)
        n_parts, stride, rows_parts = int(o['i'][4]), int(o['i'][5]), int(o['i'][6])
        if n_parts > 0 and int(o['r'][3]['buf']) >= 0:
            assert N == ld
            parts = self.tail(o['r'][3], np.float32)
            jj = np.arange(rows_parts)[:, None] * N + np.arange(N)[None, :]
            acc = X[ii[:rows_parts]].astype(np.float32)
            for p_ in range(n_parts):
                acc = acc + parts[p_ * stride + jj]
            X[ii[:rows_parts]] = acc
        if dact == L.DACT_RELU:
            X[ii] = np.where(z > 0, X[ii], 0.0)
        else:
            X[ii] = X[ii] * (0.5 * (1 + erf(z / math.sqrt(2))) + z * np.exp(-0.5 * z * z) / math.sqrt(2 * math.pi))
        amax = self.tail(o['r'][2], np.float32)
        if amax is not None:
            amax[0] = max(float(amax[0]), float(np.abs(X[ii]).max()))

    def op_transpose32(self, o, problems):
        rows, cols, ld_s, ld_d, batch, sb, db = (int(v) for v in o['i'][:7])
        src = self.tail(o['r'][1], np.float32)
        dst = self.tail(o['r'][0], np.float32)
        for b in range(batch):
            S = np.lib.stride_tricks.as_strided(src[b * sb:], shape=(rows, cols), strides=(4 * ld_s, 4))
            D = np.lib.stride_tricks.as_strided(dst[b * db:], shape=(cols, rows), strides=(4 * ld_d, 4))
            D[:] = S.T

    def op_add(self, o, problems):
        n = int(o['i'][0])
        self.fview(o['r'][0], n)[:] += self.fview(o['r'][1], n)
