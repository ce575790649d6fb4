"""pyref -- second, independent restatement of the go-jpeg2000 hot path: a slow,
literal Python transliteration written from the Go source text (not from the C
oracle), used ONLY in the build container to (a) cross-check the C oracle bit
for bit and (b) emit the committed fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Deliberately follows *different* reference entry
points than the C oracle where the reference has two equivalent forms:
  * T1 encoder: the method-per-pass form EncodeSafe (t1.go:923-947 with
    encode*PassInlined t1.go:558-914), whereas the C oracle follows EncodeFast5
    (t1_fast5.go).  Agreement of the two is evidence that the reference's own two
    encoders agree (the reference asserts that only via round trips).
  * MQ table: the 94-entry literal semantics are reproduced from the ISO 47-row
    table here too, but built independently (see _mq_tables).
  * sign context: computed by the branchy rule (t1.go:387-460), not the LUT.

Go semantics: int32 wraparound via _i32(), uint32 via & M32, Go shifts >= width
give 0 (Python big ints make that explicit), int32(float64) == math.trunc.
"""
import math

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def _i32(v):
    v &= M32
    return v - (1 << 32) if v & 0x80000000 else v


# ------------------------------------------------------------------ mct.go ---
def dc_shift_forward(data, precision):          # mct.go:96-101
    s = 1 << (precision - 1)
    return [_i32(v - s) for v in data]


def dc_shift_inverse(data, precision):          # mct.go:113-118
    s = 1 << (precision - 1)
    return [_i32(v + s) for v in data]


def forward_rct(r, g, b):                       # mct.go:28-38
    Y, U, V = [], [], []
    for i in range(len(r)):
        Y.append(_i32(r[i] + 2 * g[i] + b[i]) >> 2)
        U.append(_i32(b[i] - g[i]))
        V.append(_i32(r[i] - g[i]))
    return Y, U, V


def inverse_rct(y, u, v):                       # mct.go:56-66
    R, G, B = [], [], []
    for i in range(len(y)):
        g = _i32(y[i] - (_i32(u[i] + v[i]) >> 2))
        R.append(_i32(v[i] + g)); G.append(g); B.append(_i32(u[i] + g))
    return R, G, B


def forward_ict(r, g, b):                       # mct.go:14-24
    Y, Cb, Cr = [], [], []
    for i in range(len(r)):
        Y.append(0.299 * r[i] + 0.587 * g[i] + 0.114 * b[i])
        Cb.append(-0.16875 * r[i] - 0.33126 * g[i] + 0.5 * b[i])
        Cr.append(0.5 * r[i] - 0.41869 * g[i] - 0.08131 * b[i])
    return Y, Cb, Cr


def inverse_ict(y, cb, cr):                     # mct.go:43-53
    R, G, B = [], [], []
    for i in range(len(y)):
        R.append(y[i] + 1.402 * cr[i])
        G.append(y[i] - 0.34413 * cb[i] - 0.71414 * cr[i])
        B.append(y[i] + 1.772 * cb[i])
    return R, G, B


# ------------------------------------------------------------------ dwt.go ---
def _deinterleave(d, n):                        # dwt.go:265-284
    half = (n + 1) // 2
    tmp = [0] * n
    j = 0
    for i in range(0, n, 2):
        tmp[j] = d[i]; j += 1
    j = half
    for i in range(1, n, 2):
        tmp[j] = d[i]; j += 1
    d[:n] = tmp


def _interleave(d, n):                          # dwt.go:287-306
    tmp = list(d[:n])
    half = (n + 1) // 2
    i = 0
    for j in range(half):
        d[i] = tmp[j]; i += 2
    i = 1
    for j in range(half, n):
        d[i] = tmp[j]; i += 2


def forward53(d, n):                            # dwt.go:73-118 (d: list, in place)
    if n < 2:
        return
    for i in range(1, n - 1, 2):
        d[i] = _i32(d[i] - (_i32(d[i - 1] + d[i + 1]) >> 1))
    if n & 1 == 0:
        d[n - 1] = _i32(d[n - 1] - d[n - 2])
    d[0] = _i32(d[0] + (_i32(d[1] + d[1] + 2) >> 2))
    for i in range(2, n - 1, 2):
        d[i] = _i32(d[i] + (_i32(d[i - 1] + d[i + 1] + 2) >> 2))
    if n & 1 != 0:
        d[n - 1] = _i32(d[n - 1] + (_i32(d[n - 2] + d[n - 2] + 2) >> 2))
    _deinterleave(d, n)


def inverse53(d, n):                            # dwt.go:122-147
    if n < 2:
        return
    _interleave(d, n)
    d[0] = _i32(d[0] - (_i32(d[1] + d[1] + 2) >> 2))
    for i in range(2, n - 1, 2):
        d[i] = _i32(d[i] - (_i32(d[i - 1] + d[i + 1] + 2) >> 2))
    if n & 1 != 0:
        d[n - 1] = _i32(d[n - 1] - (_i32(d[n - 2] + d[n - 2] + 2) >> 2))
    for i in range(1, n - 1, 2):
        d[i] = _i32(d[i] + (_i32(d[i - 1] + d[i + 1]) >> 1))
    if n & 1 == 0:
        d[n - 1] = _i32(d[n - 1] + d[n - 2])


ALPHA97 = -1.586134342059924                    # dwt.go:150-157
BETA97 = -0.052980118572961
GAMMA97 = 0.882911075530934
DELTA97 = 0.443506852043971
K97 = 1.230174104914001
K97INV = 0.812893066115961


def forward97(d, n):                            # dwt.go:161-210
    if n < 2:
        return
    for c_odd, c_even in ((ALPHA97, BETA97), (GAMMA97, DELTA97)):
        for i in range(1, n - 1, 2):
            d[i] += c_odd * (d[i - 1] + d[i + 1])
        if n & 1 == 0:
            d[n - 1] += 2 * c_odd * d[n - 2]
        d[0] += 2 * c_even * d[1]
        for i in range(2, n - 1, 2):
            d[i] += c_even * (d[i - 1] + d[i + 1])
        if n & 1 != 0:
            d[n - 1] += 2 * c_even * d[n - 2]
    for i in range(0, n, 2):
        d[i] *= K97INV
    for i in range(1, n, 2):
        d[i] *= K97
    _deinterleave(d, n)


def inverse97(d, n):                            # dwt.go:213-262
    if n < 2:
        return
    _interleave(d, n)
    for i in range(0, n, 2):
        d[i] *= K97
    for i in range(1, n, 2):
        d[i] *= K97INV
    for c_even, c_odd in ((DELTA97, GAMMA97), (BETA97, ALPHA97)):
        d[0] -= 2 * c_even * d[1]
        for i in range(2, n - 1, 2):
            d[i] -= c_even * (d[i - 1] + d[i + 1])
        if n & 1 != 0:
            d[n - 1] -= 2 * c_even * d[n - 2]
        for i in range(1, n - 1, 2):
            d[i] -= c_odd * (d[i - 1] + d[i + 1])
        if n & 1 == 0:
            d[n - 1] -= 2 * c_odd * d[n - 2]


def _fwd2d(d, w, h, f1d):                       # dwt.go:356-407 / 432-451
    for y in range(h):
        row = d[y * w:(y + 1) * w]
        f1d(row, w)
        d[y * w:(y + 1) * w] = row
    for x in range(w):
        col = [d[y * w + x] for y in range(h)]
        f1d(col, h)
        for y in range(h):
            d[y * w + x] = col[y]


def _inv2d(d, w, h, i1d):                       # dwt.go:410-429 / 454-473
    for x in range(w):
        col = [d[y * w + x] for y in range(h)]
        i1d(col, h)
        for y in range(h):
            d[y * w + x] = col[y]
    for y in range(h):
        row = d[y * w:(y + 1) * w]
        i1d(row, w)
        d[y * w:(y + 1) * w] = row


def forward2d53(d, w, h): _fwd2d(d, w, h, forward53)
def inverse2d53(d, w, h): _inv2d(d, w, h, inverse53)
def forward2d97(d, w, h): _fwd2d(d, w, h, forward97)
def inverse2d97(d, w, h): _inv2d(d, w, h, inverse97)


def _decompose(d, w, h, levels, f2d):           # dwt.go:524-531 / 551-558 (prefix layout)
    for _ in range(levels):
        pre = d[:w * h]
        f2d(pre, w, h)
        d[:w * h] = pre
        w = (w + 1) // 2; h = (h + 1) // 2


def _reconstruct(d, w, h, levels, i2d):         # dwt.go:534-548 / 561-573
    dims = []
    for _ in range(levels):
        dims.append((w, h)); w = (w + 1) // 2; h = (h + 1) // 2
    for (lw, lh) in reversed(dims):
        pre = d[:lw * lh]
        i2d(pre, lw, lh)
        d[:lw * lh] = pre


def decompose53(d, w, h, levels): _decompose(d, w, h, levels, forward2d53)
def reconstruct53(d, w, h, levels): _reconstruct(d, w, h, levels, inverse2d53)
def decompose97(d, w, h, levels): _decompose(d, w, h, levels, forward2d97)
def reconstruct97(d, w, h, levels): _reconstruct(d, w, h, levels, inverse2d97)


def _trunc32(v):
    return _i32(int(math.trunc(v)))


def preprocess(planes, w, h, precision, lossless, num_resolutions, quality=0):  # encoder.go:216-281
    planes = [dc_shift_forward(list(p), precision) for p in planes]
    if len(planes) >= 3:
        if lossless:
            planes[0], planes[1], planes[2] = forward_rct(planes[0], planes[1], planes[2])
        else:
            f = forward_ict(*[[float(v) for v in planes[c]] for c in range(3)])
            for c in range(3):
                planes[c] = [_trunc32(v + 0.5) if v >= 0 else _trunc32(v - 0.5) for v in f[c]]
    levels = num_resolutions - 1
    if levels <= 0:
        levels = 5
    out = []
    for p in planes:
        if lossless:
            decompose53(p, w, h, levels)
            out.append(p)
        else:
            f = [float(v) for v in p]
            decompose97(f, w, h, levels)
            q = quality if quality > 0 else 100
            step = 1.0 / float(q)
            out.append([_trunc32(v / step + 0.5) if v >= 0 else _trunc32(v / step - 0.5) for v in f])
    return out


# ------------------------------------------------------------------ mqc.go ---
def _mq_tables():
    """94-state table in the reference's interleaved form (mqc.go:21-116), built from
    ISO/IEC 15444-1 Table C.2 rows (Qe, NMPS, NLPS, SWITCH)."""
    rows = """5601 1 1 1|3401 2 6 0|1801 3 9 0|0AC1 4 12 0|0521 5 29 0|0221 38 33 0|5601 7 6 1|
    5401 8 14 0|4801 9 14 0|3801 10 14 0|3001 11 17 0|2401 12 18 0|1C01 13 20 0|1601 29 21 0|
    5601 15 14 1|5401 16 14 0|5101 17 15 0|4801 18 16 0|3801 19 17 0|3401 20 18 0|3001 21 19 0|
    2801 22 19 0|2401 23 20 0|2201 24 21 0|1C01 25 22 0|1801 26 23 0|1601 27 24 0|1401 28 25 0|
    1201 29 26 0|1101 30 27 0|0AC1 31 28 0|09C1 32 29 0|08A1 33 30 0|0521 34 31 0|0441 35 32 0|
    02A1 36 33 0|0221 37 34 0|0141 38 35 0|0111 39 36 0|0085 40 37 0|0049 41 38 0|0025 42 39 0|
    0015 43 40 0|0009 44 41 0|0005 45 42 0|0001 45 43 0|5601 46 46 0"""
    qe, nmps, nlps = [], [], []
    for row in rows.replace("\n", "").split("|"):
        q, nm, nl, sw = row.split()
        q = int(q, 16); nm = int(nm); nl = int(nl); sw = int(sw)
        for mps in (0, 1):
            qe.append(q)
            nmps.append(2 * nm + mps)
            nlps.append(2 * nl + ((1 - mps) if sw else mps))
    return qe, nmps, nlps


MQ_QE, MQ_NMPS, MQ_NLPS = _mq_tables()
CTX_ZC0, CTX_SC0, CTX_MAG0, CTX_MAG1, CTX_MAG2, CTX_RL, CTX_UNI, NUM_CTX = 0, 9, 14, 15, 16, 17, 18, 19
BAND_LL, BAND_HL, BAND_LH, BAND_HH = 0, 1, 2, 3


class MQEncoder:                                # mqc.go:169-349
    def __init__(self):
        self.A = 0x8000; self.C = 0; self.CT = 12
        self.buf = bytearray(1); self.bp = 0
        self.contexts = [0] * NUM_CTX
        self.contexts[CTX_UNI] = 92

    def encode(self, ctx, decision):            # mqc.go:224-255
        st = self.contexts[ctx]; qe = MQ_QE[st]; mps = st & 1
        self.A = (self.A - qe) & M32
        if decision == mps:
            if self.A & 0x8000 == 0:
                if self.A < qe:
                    self.A = qe
                else:
                    self.C = (self.C + qe) & M32
                self.contexts[ctx] = MQ_NMPS[st]
                self._renorm()
            else:
                self.C = (self.C + qe) & M32
        else:
            if self.A < qe:
                self.C = (self.C + qe) & M32
            else:
                self.A = qe
            self.contexts[ctx] = MQ_NLPS[st]
            self._renorm()

    def _renorm(self):                          # mqc.go:258-267
        while self.A & 0x8000 == 0:
            self.A = (self.A << 1) & M32; self.C = (self.C << 1) & M32
            self.CT -= 1
            if self.CT == 0:
                self._byte_out()

    def _put(self, v):
        self.bp += 1
        if self.bp >= len(self.buf):
            self.buf.append(0)
        self.buf[self.bp] = v & 0xFF

    def _byte_out(self):                        # mqc.go:270-310
        if self.buf[self.bp] == 0xFF:
            self._put(self.C >> 20); self.C &= 0xFFFFF; self.CT = 7
        elif self.C & 0x8000000 == 0:
            self._put(self.C >> 19); self.C &= 0x7FFFF; self.CT = 8
        else:
            self.buf[self.bp] += 1
            if self.buf[self.bp] == 0xFF:
                self.C &= 0x7FFFFFF
                self._put(self.C >> 20); self.C &= 0xFFFFF; self.CT = 7
            else:
                self._put(self.C >> 19); self.C &= 0x7FFFF; self.CT = 8

    def flush(self):                            # mqc.go:313-341
        temp = (self.C + self.A) & M32
        self.C |= 0xFFFF
        if self.C >= temp:
            self.C = (self.C - 0x8000) & M32
        self.C = (self.C << self.CT) & M32; self._byte_out()
        self.C = (self.C << self.CT) & M32; self._byte_out()
        end = self.bp + 1
        if end > 0 and self.buf[end - 1] == 0xFF:
            end -= 1
        return bytes(self.buf[1:end]) if end > 1 else b""


class MQDecoder:                                # mqc.go:352-497
    def __init__(self, data):
        self.data = bytes(data)
        self.A = 0x8000; self.C = 0; self.CT = 0; self.bp = -1
        self.contexts = [0] * NUM_CTX
        self.contexts[CTX_UNI] = 92
        if len(self.data) == 0:
            self.C = 0xFF << 16
        else:
            self.bp = 0; self.C = self.data[0] << 16
        self._byte_in()
        self.C = (self.C << 7) & M32
        self.CT = (self.CT - 7) & M32
        self.A = 0x8000

    def _byte_in(self):                         # mqc.go:402-439
        if self.bp < 0:
            self.bp = 0
        if self.bp >= len(self.data):
            self.C = (self.C + 0xFF00) & M32; self.CT = 8
            return
        nxt = self.data[self.bp + 1] if self.bp + 1 < len(self.data) else 0xFF
        if self.data[self.bp] == 0xFF:
            if nxt > 0x8F:
                self.C = (self.C + 0xFF00) & M32; self.CT = 8
            else:
                self.bp += 1; self.C = (self.C + (nxt << 9)) & M32; self.CT = 7
        else:
            self.bp += 1; self.C = (self.C + (nxt << 8)) & M32; self.CT = 8

    def _renorm(self):                          # mqc.go:488-497
        while self.A & 0x8000 == 0:
            if self.CT == 0:
                self._byte_in()
            self.A = (self.A << 1) & M32; self.C = (self.C << 1) & M32
            self.CT = (self.CT - 1) & M32

    def decode(self, ctx):                      # mqc.go:443-485
        st = self.contexts[ctx]; qe = MQ_QE[st]; mps = st & 1
        self.A = (self.A - qe) & M32
        if (self.C >> 16) < qe:
            if self.A < qe:
                self.A = qe; d = mps; self.contexts[ctx] = MQ_NMPS[st]
            else:
                self.A = qe; d = 1 - mps; self.contexts[ctx] = MQ_NLPS[st]
            self._renorm()
            return d
        self.C = (self.C - (qe << 16)) & M32
        if self.A & 0x8000 == 0:
            if self.A < qe:
                d = 1 - mps; self.contexts[ctx] = MQ_NLPS[st]
            else:
                d = mps; self.contexts[ctx] = MQ_NMPS[st]
            self._renorm()
            return d
        return mps


# ------------------------------------------------------------------- t1.go ---
T1_SIG, T1_VISIT, T1_REFINE, T1_SIGN_NEG = 1, 2, 4, 8


def zc_context(band, w, e, n, s, nw, ne, sw, se):   # t1_luts.go:34-110 rule form
    h = w + e; v = n + s; d = nw + ne + sw + se
    if band == BAND_HL:
        h, v = v, h
    if band == BAND_HH:
        hv = h + v
        if hv >= 3: return 8
        if hv == 2: return 7 if d >= 2 else (6 if d >= 1 else 5)
        if hv == 1: return 4 if d >= 2 else 3
        return 2 if d >= 2 else (1 if d >= 1 else 0)
    if h == 2: return 8
    if h == 1: return 7 if v >= 1 else (6 if d >= 1 else 5)
    if v == 2: return 4
    if v == 1: return 3 if d >= 1 else 2
    return 1 if d >= 2 else 0


class T1:
    """Straight transliteration of the method-per-pass T1 (t1.go)."""

    def __init__(self, w, h):
        self.w = w; self.h = h; self.stride = w + 2
        self.data = [0] * (w * h)
        self.flags = [0] * ((w + 2) * (h + 2))

    def fi(self, x, y): return (y + 1) * self.stride + x + 1     # t1.go:308-310

    def set_data(self, data):                   # t1.go:292-304
        for i, v in enumerate(data):
            if v < 0:
                self.data[i] = _i32(-v)
                self.flags[self.fi(i % self.w, i // self.w)] |= T1_SIGN_NEG
            else:
                self.data[i] = v

    def sig(self, x, y): return 1 if self.flags[self.fi(x, y)] & T1_SIG else 0

    def neg(self, x, y): return 1 if self.flags[self.fi(x, y)] & T1_SIGN_NEG else 0

    def zc(self, x, y):                         # t1.go:349-384
        s = self.sig
        return zc_context(self.band, s(x - 1, y), s(x + 1, y), s(x, y - 1), s(x, y + 1),
                          s(x - 1, y - 1), s(x + 1, y - 1), s(x - 1, y + 1), s(x + 1, y + 1))

    def has_sig_neighbor(self, x, y):           # t1.go:1087-1092
        s = self.sig
        return bool(s(x - 1, y) | s(x + 1, y) | s(x, y - 1) | s(x, y + 1) | s(x - 1, y - 1) |
                    s(x + 1, y - 1) | s(x - 1, y + 1) | s(x + 1, y + 1))

    def sc(self, x, y):                         # t1.go:387-460
        def contrib(xx, yy):
            if not self.sig(xx, yy):
                return 0
            return -1 if self.neg(xx, yy) else 1
        hc = contrib(x - 1, y) + contrib(x + 1, y)
        vc = contrib(x, y - 1) + contrib(x, y + 1)
        pred = 0
        if hc < 0:
            pred = 1; hc = -hc
        if hc == 0:
            if vc < 0:
                pred = 1; vc = -vc
        ctx = 0
        if hc == 1:
            ctx = 4 if vc == 1 else (2 if vc == 0 else 1)
        elif hc == 0:
            ctx = 1 if vc == 1 else 0
        elif hc == 2:
            ctx = 3
        return CTX_SC0 + ctx, pred

    def mr(self, x, y):                         # t1.go:463-479
        if self.flags[self.fi(x, y)] & T1_REFINE == 0:
            return CTX_MAG1 if self.has_sig_neighbor(x, y) else CTX_MAG0
        return CTX_MAG2

    def can_rl(self, x, y):                     # t1.go:774-813 == 1195-1208
        if y + 4 > self.h:
            return False
        for yy in range(y, y + 4):
            if self.flags[self.fi(x, yy)] & (T1_SIG | T1_VISIT):
                return False
            if self.has_sig_neighbor(x, yy):
                return False
        return True

    # ---- encoder (EncodeSafe, t1.go:923-947) ----
    def encode(self, band):
        self.band = band
        mq = MQEncoder()
        maxval = max(self.data) if self.data else 0
        if maxval <= 0:
            return b"", 0
        numbps = int(math.ceil(math.log2(float(maxval + 1))))    # t1.go:937
        w, h = self.w, self.h

        def code_sign(x, y):
            ctx, pred = self.sc(x, y)
            mq.encode(ctx, self.neg(x, y) ^ pred)

        def newly_sig(x, y):
            self.flags[self.fi(x, y)] |= T1_SIG

        for bp in range(numbps - 1, -1, -1):
            bit = 1 << bp
            for y in range(h):                  # t1.go:558-639
                for x in range(w):
                    i = self.fi(x, y)
                    if self.flags[i] & T1_SIG:
                        continue
                    if not self.has_sig_neighbor(x, y):
                        continue
                    sig = 1 if self.data[y * w + x] & bit else 0
                    mq.encode(self.zc(x, y), sig)
                    if sig:
                        code_sign(x, y); newly_sig(x, y)
                    self.flags[i] |= T1_VISIT
            for y in range(h):                  # t1.go:642-683
                for x in range(w):
                    i = self.fi(x, y)
                    f = self.flags[i]
                    if f & T1_SIG == 0 or f & T1_VISIT != 0:
                        continue
                    mq.encode(self.mr(x, y), 1 if self.data[y * w + x] & bit else 0)
                    self.flags[i] |= T1_REFINE
            for y in range(0, h, 4):            # t1.go:686-770
                for x in range(w):
                    if self.can_rl(x, y):       # t1.go:816-914
                        first = -1
                        for i in range(4):
                            if y + i >= h:
                                break
                            if self.data[(y + i) * w + x] & bit:
                                first = i; break
                        if first == -1:
                            mq.encode(CTX_RL, 0)
                            continue
                        mq.encode(CTX_RL, 1)
                        mq.encode(CTX_UNI, (first >> 1) & 1)
                        mq.encode(CTX_UNI, first & 1)
                        code_sign(x, y + first); newly_sig(x, y + first)
                        for i in range(first + 1, 4):
                            if y + i >= h:
                                break
                            sig = 1 if self.data[(y + i) * w + x] & bit else 0
                            mq.encode(self.zc(x, y + i), sig)
                            if sig:
                                code_sign(x, y + i); newly_sig(x, y + i)
                        continue
                    for yy in range(y, min(y + 4, h)):
                        i = self.fi(x, yy)
                        f = self.flags[i]
                        if f & T1_VISIT:
                            self.flags[i] &= ~T1_VISIT
                            continue
                        if f & T1_SIG:
                            continue
                        sig = 1 if self.data[yy * w + x] & bit else 0
                        mq.encode(self.zc(x, yy), sig)
                        if sig:
                            code_sign(x, yy); newly_sig(x, yy)
        return mq.flush(), numbps

    # ---- decoder (t1.go:1261-1410) ----
    def decode(self, data, numbps, band):
        self.band = band
        mq = MQDecoder(data)
        w, h = self.w, self.h
        self.data = [0] * (w * h)
        self.flags = [0] * len(self.flags)

        def dec_sign(x, y):
            ctx, pred = self.sc(x, y)
            if mq.decode(ctx) ^ pred:
                self.flags[self.fi(x, y)] |= T1_SIGN_NEG

        for bp in range(numbps - 1, -1, -1):
            bit = _i32(1 << bp) if bp < 32 else 0
            for y in range(h):
                for x in range(w):
                    i = self.fi(x, y)
                    if self.flags[i] & T1_SIG or not self.has_sig_neighbor(x, y):
                        continue
                    if mq.decode(self.zc(x, y)):
                        self.data[y * w + x] = bit
                        dec_sign(x, y); self.flags[i] |= T1_SIG
                    self.flags[i] |= T1_VISIT
            for y in range(h):
                for x in range(w):
                    i = self.fi(x, y)
                    if self.flags[i] & T1_SIG == 0 or self.flags[i] & T1_VISIT:
                        continue
                    if mq.decode(self.mr(x, y)):
                        self.data[y * w + x] |= bit
                    self.flags[i] |= T1_REFINE
            for y in range(0, h, 4):
                for x in range(w):
                    if self.can_rl(x, y):
                        if mq.decode(CTX_RL) == 0:
                            continue
                        pos = mq.decode(CTX_UNI) << 1
                        pos |= mq.decode(CTX_UNI)
                        self.data[(y + pos) * w + x] = bit
                        dec_sign(x, y + pos); self.flags[self.fi(x, y + pos)] |= T1_SIG
                        for i in range(pos + 1, 4):
                            if y + i >= h:
                                break
                            if mq.decode(self.zc(x, y + i)):
                                self.data[(y + i) * w + x] = bit
                                dec_sign(x, y + i); self.flags[self.fi(x, y + i)] |= T1_SIG
                        continue
                    for yy in range(y, min(y + 4, h)):
                        i = self.fi(x, yy)
                        if self.flags[i] & T1_VISIT:
                            self.flags[i] &= ~T1_VISIT
                            continue
                        if self.flags[i] & T1_SIG:
                            continue
                        if mq.decode(self.zc(x, yy)):
                            self.data[yy * w + x] = bit
                            dec_sign(x, yy); self.flags[i] |= T1_SIG
        out = []
        for i, v in enumerate(self.data):
            out.append(_i32(-v) if self.flags[self.fi(i % w, i // w)] & T1_SIGN_NEG else v)
        return out


def t1_encode(data, w, h, band):
    t = T1(w, h); t.set_data(list(data))
    return t.encode(band)


def t1_decode(data, numbps, band, w, h):
    return T1(w, h).decode(data, numbps, band)


# ------------------------------------------------------------------- ht.go ---
def load_ht_tables(path):
    """Parse the generated numeric header (tools/gen_ht_tables.py output)."""
    import re
    txt = open(path).read()
    def tab(name):
        m = re.search(r"#define %s_INIT \{(.*?)\n\}" % name, txt, re.S)
        return [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{4}", m.group(1))]
    return tab("J2K_HT_VLC_TBL0"), tab("J2K_HT_VLC_TBL1")


class GoPanic(Exception):
    pass


class HTEncoder:                                # ht.go:872-1391
    def __init__(self, w, h, tables):
        self.w = w; self.h = h; self.tbl0, self.tbl1 = tables

    def _vlc_write(self, val, nbits):           # ht.go:1266-1286
        self.vtmp |= (val << self.vbits) & M64
        self.vbits += nbits
        while self.vbits >= 8:
            b = self.vtmp & 0xFF
            if self.vlast > 0x8F and (b & 0x7F) == 0x7F:
                b &= 0x7F
            if self.vpos < 0:
                raise GoPanic("vlc index out of range")
            self.vdata[self.vpos] = b; self.vpos -= 1; self.vlast = b
            self.vtmp >>= 8; self.vbits -= 8

    def _ms_write(self, val, nbits):            # ht.go:1303-1327
        self.mtmp |= (val << self.mbits) & M64
        self.mbits += nbits
        while self.mbits >= 8:
            b = self.mtmp & 0xFF
            if self.mpos >= len(self.mdata):
                raise GoPanic("magsgn index out of range")
            if self.mlast == 0xFF:
                b &= 0x7F
                self.mdata[self.mpos] = b; self.mpos += 1
                self.mtmp >>= 7; self.mbits -= 7
            else:
                self.mdata[self.mpos] = b; self.mpos += 1
                self.mtmp >>= 8; self.mbits -= 8
            self.mlast = b

    def _vlc_quad(self, context, rho, initial):  # ht.go:1199-1226
        tbl = self.tbl0 if initial else self.tbl1
        for cwd in range(128):
            e = tbl[(context << 7) | cwd]
            if ((e >> 4) & 0xF) == rho and (e & 0xF) > 0:
                self._vlc_write(cwd, e & 0xF)
                return
        self._vlc_write(0, 1)

    def _uvlc(self, mode, u1, u2):              # ht.go:1229-1263
        def one(u):
            if u <= 1: self._vlc_write(1, 1)
            elif u <= 2: self._vlc_write(2, 2)
            else:
                self._vlc_write(0, 3); self._vlc_write((u - 3) & M32, 5)
        if mode in (1, 2):
            one(u1 if mode == 1 else u2)
        elif mode == 3:
            one(u1); one(u2)

    def encode(self, data):                     # ht.go:942-1045
        w, h = self.w, self.h
        maxmag = 0
        for v in data:
            if v < 0:
                v = _i32(-v)
            if v > maxmag:
                maxmag = v
        if maxmag == 0:
            return b""
        max_size = max(w * h * 2, 64)
        mel_len = max_size // 4
        self.vdata = bytearray(max_size // 2); self.vpos = len(self.vdata) - 1
        self.vtmp = 0; self.vbits = 0; self.vlast = 0
        self.mdata = bytearray(max_size // 2); self.mpos = 0
        self.mtmp = 0; self.mbits = 0; self.mlast = 0
        qcols = (w + 3) // 4
        sigma1 = [0] * (qcols + 1)
        for y in range(0, h, 4):                # ht.go:1054-1195
            initial = (y == 0)
            for qx in range(0, qcols, 2):
                def quad(q):
                    return [(data[y * w + q * 4 + i]) for i in range(4) if q * 4 + i < w]
                q1, q2 = quad(qx), quad(qx + 1)
                rho = sum((1 << i) for i, v in enumerate(q1) if v != 0)
                rho2 = sum((1 << i) for i, v in enumerate(q2) if v != 0)
                if initial:
                    context = (sigma1[qx - 1] >> 4) if qx > 0 else 0
                else:
                    context = sigma1[qx] >> 4
                self._vlc_quad(context, rho, initial)
                sigma1[qx] = rho
                context2 = (rho >> 2) | (sigma1[qx + 1] >> 4)
                self._vlc_quad(context2, rho2, initial)
                sigma1[qx + 1] = rho2
                if rho != 0 or rho2 != 0:
                    def uval(q):
                        u = 1
                        for v in q:
                            if v < 0:
                                v = _i32(-v)
                            if (v & M32) >= ((1 << u) & M32):
                                u += 1
                        return u
                    self._uvlc((1 if rho else 0) | (2 if rho2 else 0), uval(q1), uval(q2))
                for q, r in ((q1, rho), (q2, rho2)):
                    for i, v in enumerate(q):
                        if not r & (1 << i):
                            continue
                        sign = 0
                        if v < 0:
                            sign = 1; v = _i32(-v)
                        mag = v & M32
                        if mag >= 0x80000000:
                            raise GoPanic("emb loop does not terminate")
                        emb = 1
                        while mag >= ((1 << emb) & M32):
                            emb += 1
                        self._ms_write(mag & (((1 << (emb - 1)) - 1) & M32), emb - 1)
                        self._ms_write(sign, 1)
        while self.vbits > 0:                   # vlcFlush ht.go:1289-1300
            if self.vpos < 0:
                raise GoPanic("vlc index out of range")
            self.vdata[self.vpos] = self.vtmp & 0xFF; self.vpos -= 1
            self.vtmp >>= 8; self.vbits = max(self.vbits - 8, 0)
        while self.mbits > 0:                   # magSgnFlush ht.go:1330-1341
            if self.mpos >= len(self.mdata):
                raise GoPanic("magsgn index out of range")
            self.mdata[self.mpos] = self.mtmp & 0xFF; self.mpos += 1
            self.mtmp >>= 8; self.mbits = max(self.mbits - 8, 0)
        vlc_len = len(self.vdata) - self.vpos - 1
        scup = mel_len + vlc_len + 2
        out = bytearray(self.mdata[:self.mpos]) + bytearray(mel_len)
        for i in range(vlc_len):
            out.append(self.vdata[len(self.vdata) - 1 - i])
        out += bytes([(scup >> 8) & 0xFF, scup & 0xFF])
        return bytes(out)


class _Rev:
    pass


class HTDecoder:                                # ht.go:14-864
    def __init__(self, w, h, tables):
        self.w = w; self.h = h; self.tbl0, self.tbl1 = tables

    # -- reverse (VLC) stream --
    def _rev_read(self):                        # ht.go:317-378
        v = self.v
        if v.bits > 32:
            return
        val = 0
        if v.size > 3:
            p = v.pos - 3
            if p >= 0 and p + 3 < len(v.data):
                val = v.data[p] | v.data[p + 1] << 8 | v.data[p + 2] << 16 | v.data[p + 3] << 24
            v.pos -= 4; v.size -= 4
        elif v.size > 0:
            i = 24
            while v.size > 0:
                if 0 <= v.pos < len(v.data):
                    val |= v.data[v.pos] << i
                    v.pos -= 1
                v.size -= 1; i -= 8
        tmp = val >> 24
        bits = 7 if (v.unstuff and ((val >> 24) & 0x7F) == 0x7F) else 8
        unstuff = (val >> 24) > 0x8F
        for sh in (16, 8, 0):
            byte = (val >> sh) & 0xFF
            tmp |= (byte << bits) & M32
            bits += 7 if (unstuff and (byte & 0x7F) == 0x7F) else 8
            unstuff = byte > 0x8F
        v.unstuff = unstuff
        v.tmp = (v.tmp | (tmp << v.bits)) & M64
        v.bits = (v.bits + bits) & M32

    def _rev_fetch(self):
        if self.v.bits < 32:
            self._rev_read()
            if self.v.bits < 32:
                self._rev_read()
        return self.v.tmp & M32

    def _rev_adv(self, n):
        self.v.tmp >>= n; self.v.bits = (self.v.bits - n) & M32

    # -- forward (MagSgn) stream --
    def _fwd_read(self):                        # ht.go:432-501
        f = self.f
        if f.bits > 32:
            return
        val = 0
        if f.size > 3:
            if f.pos + 3 < len(f.data):
                val = f.data[f.pos] | f.data[f.pos + 1] << 8 | f.data[f.pos + 2] << 16 | f.data[f.pos + 3] << 24
            f.pos += 4; f.size -= 4
        elif f.size > 0:
            if f.x != 0:
                val = M32
            i = 0
            while f.size > 0:
                if f.pos < len(f.data):
                    val = (val & (~(0xFF << i) & M32)) | (f.data[f.pos] << i)
                    f.pos += 1
                f.size -= 1; i += 8
        else:
            if f.x != 0:
                val = M32
        bits = 7 if f.unstuff else 8
        t = val & 0xFF
        unstuff = (val & 0xFF) == 0xFF
        for sh in (8, 16, 24):
            byte = (val >> sh) & 0xFF
            t |= (byte << bits) & M32
            bits += 7 if unstuff else 8
            unstuff = byte == 0xFF
        f.unstuff = unstuff
        f.tmp = (f.tmp | (t << f.bits)) & M64
        f.bits = (f.bits + bits) & M32

    def _fwd_fetch(self):
        if self.f.bits < 32:
            self._fwd_read()
            if self.f.bits < 32:
                self._fwd_read()
        return self.f.tmp & M32

    def _fwd_adv(self, n):
        self.f.tmp >>= n; self.f.bits = (self.f.bits - n) & M32

    def _init_mel(self, data, lcup, scup):      # ht.go:153-195
        pos = lcup - scup; size = scup - 1; unstuff = False
        num = min(4 - (pos & 3), 4)
        i = 0
        while i < num and size > 0:
            if unstuff and pos < len(data) and data[pos] > 0x8F:
                return False
            if size > 0 and pos < len(data):
                b = data[pos]; pos += 1; size -= 1
            else:
                b = 0xFF
            if size == 1:
                b |= 0x0F
            unstuff = (b == 0xFF)
            i += 1
        return True

    _DEC = [3 | (5 << 2) | (5 << 5), 1 | (1 << 5), 2 | (2 << 5), 1 | (1 << 5),
            3 | (1 << 2) | (3 << 5), 1 | (1 << 5), 2 | (2 << 5), 1 | (1 << 5)]

    def _uvlc(self, vlc, mode, initial):        # ht.go:716-864
        dec = self._DEC
        consumed = 0
        u = [1, 1]
        def suffix(t, vlc):
            sl = (t >> 2) & 7
            return sl, ((t >> 5) + (vlc & (((1 << sl) - 1) & M32))) & M32
        if mode == 0:
            pass
        elif mode <= 2:
            t = dec[vlc & 7]
            pl = t & 3; vlc >>= pl; consumed += pl
            sl, val = suffix(t, vlc); consumed += sl
            if mode == 1: u = [(val + 1) & M32, 1]
            else: u = [1, (val + 1) & M32]
        elif mode == 3:
            t1 = dec[vlc & 7]
            pl1 = t1 & 3; vlc >>= pl1; consumed += pl1
            if initial and pl1 > 2:
                u[1] = (vlc & 1) + 2; consumed += 1; vlc >>= 1
                sl, val = suffix(t1, vlc); consumed += sl
                u[0] = (val + 1) & M32
            else:
                t2 = dec[vlc & 7]
                pl2 = t2 & 3; vlc >>= pl2; consumed += pl2
                sl1, val1 = suffix(t1, vlc); consumed += sl1
                u[0] = (val1 + 1) & M32
                vlc >>= sl1
                sl2, val2 = suffix(t2, vlc); consumed += sl2
                u[1] = (val2 + 1) & M32
        return consumed, u

    def decode(self, data):                     # ht.go:93-150
        w, h = self.w, self.h
        out = [0] * (w * h)
        data = bytes(data)
        n = len(data)
        if n < 2:
            return out
        scup = data[n - 1] + ((data[n - 2] & 0x0F) << 8)
        if scup < 2 or scup > n:
            return out
        lcup = n
        if not self._init_mel(data, lcup, scup):
            return out
        # initVLC ht.go:276-314
        v = self.v = _Rev()
        v.data = data; v.pos = lcup - 2; v.size = scup - 2; v.tmp = 0; v.bits = 0; v.unstuff = False
        if 0 <= v.pos < n:
            b = data[v.pos]; v.pos -= 1
            v.tmp = b >> 4
            v.bits = 4 - ((v.tmp & 7) >> 2)
            v.unstuff = (b | 0x0F) > 0x8F
        num = 1 + (v.pos & 3)
        if num > v.size:
            num = v.size
        for _ in range(num):
            b = 0
            if 0 <= v.pos < n:
                b = data[v.pos]; v.pos -= 1
            dbits = 7 if (v.unstuff and (b & 0x7F) == 0x7F) else 8
            v.tmp |= b << v.bits
            v.bits += dbits
            v.unstuff = b > 0x8F
        v.size -= num
        self._rev_read()
        # initMagSgn ht.go:399-429
        f = self.f = _Rev()
        f.data = data; f.pos = 0; f.size = lcup - scup; f.tmp = 0; f.bits = 0; f.unstuff = False; f.x = 0xFF
        for _ in range(4 - (f.pos & 3)):
            if f.size > 0 and f.pos < n:
                b = data[f.pos]; f.pos += 1; f.size -= 1
            else:
                b = f.x
            dbits = 7 if f.unstuff else 8
            f.tmp |= b << f.bits
            f.bits += dbits
            f.unstuff = (b == 0xFF)
        self._fwd_read()

        qcols = (w + 3) // 4
        sigma1 = [0] * (qcols + 1)
        line_state = [0] * (qcols + 1)
        for y in range(0, h, 4):                # ht.go:589-711
            initial = (y == 0)
            tbl = self.tbl0 if initial else self.tbl1
            for qx in range(0, qcols, 2):
                vlc_val = self._rev_fetch()
                if initial:
                    context = (sigma1[qx - 1] >> 4) if qx > 0 else 0
                else:
                    context = (sigma1[qx] >> 4) | (line_state[qx] >> 4)
                qinf = tbl[(context << 7) | (vlc_val & 0x7F)]
                rho = (qinf >> 4) & 0xF; uoff1 = (qinf >> 3) & 1
                self._rev_adv(qinf & 0xF)
                vlc_val = self._rev_fetch()
                context2 = (rho >> 2) | (sigma1[qx + 1] >> 4)
                qinf2 = tbl[(context2 << 7) | (vlc_val & 0x7F)]
                rho2 = (qinf2 >> 4) & 0xF; uoff2 = (qinf2 >> 3) & 1
                self._rev_adv(qinf2 & 0xF)
                sigma1[qx] = rho; sigma1[qx + 1] = rho2
                mode = (uoff1 << 1) | uoff2
                if mode > 0:
                    vlc_val = self._rev_fetch()
                    consumed, u = self._uvlc(vlc_val, mode, initial)
                    self._rev_adv(consumed)
                else:
                    u = [1, 1]
                for q, r in ((0, rho), (1, rho2)):
                    emb = u[q]
                    for i in range(4):
                        if (qx + q) * 4 + i >= w:
                            break
                        if not r & (1 << i):
                            continue
                        mag_val = self._fwd_fetch()
                        mask = ((1 << emb) - 1) & M32 if emb < 32 else M32
                        top = (1 << (emb - 1)) if emb - 1 < 32 else 0
                        mag = ((mag_val & mask) + top) & M32
                        self._fwd_adv(emb)
                        sign = self._fwd_fetch() & 1
                        self._fwd_adv(1)
                        idx = y * w + (qx + q) * 4 + i
                        if idx < len(out):
                            out[idx] = _i32(-_i32(mag)) if sign else _i32(mag)
        return out


# -------------------------------------------------------- encoder.go glue ---
def enumerate_blocks(ncomp, w, h, num_res, cbw, cbh):   # encoder.go:597-673
    if num_res <= 0:
        num_res = 6
    jobs = []
    for c in range(ncomp):
        for r in range(num_res):
            for b in range(1 if r == 0 else 3):
                band = BAND_LL if r == 0 else (BAND_HL, BAND_LH, BAND_HH)[b]
                scale = 1 << (num_res - 1 - r)
                bw = (w + scale - 1) // scale; bh = (h + scale - 1) // scale
                if r > 0:
                    bw = (bw + 1) // 2; bh = (bh + 1) // 2
                cby = 0
                while cby * cbh < bh:
                    cbx = 0
                    while cbx * cbw < bw:
                        aw = min(cbw, bw - cbx * cbw); ah = min(cbh, bh - cby * cbh)
                        jobs.append((c, r, band, cbx * cbw, cby * cbh, aw, ah))
                        cbx += 1
                    cby += 1
    return jobs


def extract_block(plane, pw, ph, job):          # encoder.go:763-795
    _, _, _, x0, y0, aw, ah = job
    out = []
    for y in range(ah):
        for x in range(aw):
            sx, sy = x0 + x, y0 + y
            out.append(plane[sy * pw + sx] if (sx < pw and sy < ph) else 0)
    return out
