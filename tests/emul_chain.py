"""Software model of csrc/mlp_chain.hip, used by the CPU tests.

It walks the packed weight blob with exactly the kernel's data flow - lane/register layout of
``v_mfma_f32_16x16x4_f32``, chunk order, k-outer accumulation, slot order of the positional
encoding, the in-register skip connection, the quad layout of the forward-mode tangents - but in
numpy on the CPU, so the packing plan and the kernel's index arithmetic can be validated against
the oracle without a GPU.  One call emulates one wave (16 columns).
"""
import numpy as np

HDR_INTS, HDR_OFFS, KT = 512, 16, 4096
LANE = np.arange(64)
G, J = LANE // 16, LANE % 16


def mfma_16x16x4(a, b, c):
    """a[64], b[64] (one VGPR each), c[64,4] -> d[64,4] with the gfx950 lane layouts:
    A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C[row=4*(l>>4)+r][col=l&15]."""
    A = np.zeros((16, 4), np.float32)
    B = np.zeros((4, 16), np.float32)
    A[J, G] = a
    B[G, J] = b
    Cm = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    d = c.copy()
    for r in range(4):
        d[:, r] += Cm[4 * G + r, J]
    return d


class Blob:
    def __init__(self, blob: np.ndarray):
        self.f = np.ascontiguousarray(blob, dtype=np.float32)
        self.hdr = self.f[:HDR_INTS].view(np.int32)
        assert self.hdr[0] == 0x4E414631
        self.nc = int(self.hdr[2])
        self.offs = self.hdr[HDR_OFFS: HDR_OFFS + self.nc + 1]
        self.aux = self.f[self.hdr[4]: self.hdr[4] + self.hdr[5]]
        self.c = 0

    def acquire(self):
        w = self.f[self.offs[self.c]: self.offs[self.c + 1]]
        self.c = (self.c + 1) % self.nc
        return w


def mma_ktile(acc, xt, w, full16):
    for T in range(16 if full16 else 14):
        a = w[T * 256: (T + 1) * 256].reshape(64, 4)
        for r in range(4):
            acc[T] = mfma_16x16x4(a[:, r], xt[:, r], acc[T])


def softplus100(z):
    en = np.exp2(np.abs(z) * np.float32(-144.269504088896340736), dtype=np.float32)
    return (np.maximum(z, 0) + np.log2(1 + en, dtype=np.float32) * np.float32(0.69314718055994530942 / 100.0)).astype(np.float32)


def softplus100_grad(z):
    en = np.exp2(np.abs(z) * np.float32(-144.269504088896340736), dtype=np.float32)
    r = 1.0 / (1.0 + en)
    return np.where(z >= 0, r, en * r).astype(np.float32)


def run_layer(X, blob, bias, nt_base, nextra, full16, act, tangent):
    """X: list of [64,4] tiles (in place).  act in {'softplus','relu','none'}."""
    acc = [np.zeros((64, 4), np.float32) for _ in range(16)]
    for c in range((nt_base + 1) // 2):
        w = blob.acquire()
        mma_ktile(acc, X[2 * c], w, full16)
        if 2 * c + 1 < nt_base:
            mma_ktile(acc, X[2 * c + 1], w[KT:], full16)
    for c in range((nextra + 1) // 2):
        w = blob.acquire()
        mma_ktile(acc, X[nt_base + 2 * c], w, full16)
        if 2 * c + 1 < nextra:
            mma_ktile(acc, X[nt_base + 2 * c + 1], w[KT:], full16)
    is_val = (LANE & 3) == 0 if tangent else np.ones(64, bool)
    for T in range(16 if full16 else 14):
        b = bias[T * 16 + G[:, None] * 4 + np.arange(4)[None, :]]
        z = acc[T] + np.where(is_val[:, None], b, 0).astype(np.float32)
        if act == "softplus":
            if tangent:
                d = softplus100_grad(z)
                d = d[LANE & ~3]                       # quad broadcast of lane 0
                X[T] = np.where(is_val[:, None], softplus100(z), d * acc[T]).astype(np.float32)
            else:
                X[T] = softplus100(z)
        elif act == "relu":
            X[T] = np.maximum(z, 0).astype(np.float32)
        else:
            X[T] = z.astype(np.float32)


def encode_slots(p, q):
    """p[64,3] point of every lane, q[64] (-1 value / 0..2 derivative coordinate) -> 3 tiles [64,4]."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    last = G == 3
    cg = np.where(G == 0, x, np.where(G == 1, y, z))
    s, c = [], []
    for i in range(5):
        co = [x, y, z][i] if i < 3 else np.zeros_like(x)
        a = np.where(last, co * np.float32(32.0) if i < 3 else 0.0, cg * np.float32(1 << i)).astype(np.float32)
        s.append(np.sin(a, dtype=np.float32)); c.append(np.cos(a, dtype=np.float32))
    m = np.zeros((64, 12), np.float32)
    val = q < 0
    own = q == G
    mc = np.zeros((64, 12), np.float32); ml = np.zeros((64, 12), np.float32)
    mc[:, 0] = np.where(val, cg, np.where(own, 1.0, 0.0))
    for k in range(5):
        f = np.float32(1 << k)
        mc[:, 1 + 2 * k] = np.where(val, s[k], np.where(own, c[k] * f, 0.0))
        mc[:, 2 + 2 * k] = np.where(val, c[k], np.where(own, -(s[k] * f), 0.0))
    for i in range(3):
        ml[:, 2 * i] = np.where(val, s[i], np.where(q == i, c[i] * np.float32(32.0), 0.0))
        ml[:, 2 * i + 1] = np.where(val, c[i], np.where(q == i, -(s[i] * np.float32(32.0)), 0.0))
    m = np.where(last[:, None], ml, mc).astype(np.float32)
    return [m[:, 4 * t: 4 * t + 4].copy() for t in range(3)]


def surface_hidden(blob, p, q, tangent):
    X = [None] * 19
    E = encode_slots(p, q)
    X[0:3] = E
    run_layer(X, blob, blob.aux[0:256], 3, 0, True, "softplus", tangent)
    rs2 = np.float32(1.41421356237309504880)
    for L in range(1, 8):
        if L == 4:
            E = encode_slots(p, q)
            for t in range(14):
                X[t] = (X[t] / rs2).astype(np.float32)
            for t in range(3):
                X[14 + t] = (E[t] / rs2).astype(np.float32)
        run_layer(X, blob, blob.aux[L * 256:(L + 1) * 256], 16, 1 if L == 4 else 0, L != 3, "softplus", tangent)
    return X


def dot_row16(X, row):
    s = np.zeros(64, np.float32)
    for t in range(16):
        wv = row[t * 16 + G[:, None] * 4 + np.arange(4)[None, :]]
        s += (X[t] * wv).sum(1)
    tot = s.reshape(4, 16).sum(0)          # over the 4 lane groups
    return tot[J]


def emul_sdf_only(blob_np, pts16, R_bg):
    """pts16 [16,3] -> sdf[16]"""
    blob = Blob(blob_np)
    p = pts16[J].astype(np.float32)
    X = surface_hidden(blob, p, np.full(64, -1), False)
    sdf = dot_row16(X, blob.aux[2048:2304]) + blob.aux[2304]
    if R_bg > 0:
        sdf = np.minimum(sdf, R_bg - np.sqrt((p ** 2).sum(1)))
    return sdf[:16]


def emul_sdf_nabla(blob_np, pts4, R_bg):
    """pts4 [4,3] -> sdf[4], nabla[4,3], h7[4,256]"""
    blob = Blob(blob_np)
    p = pts4[J >> 2].astype(np.float32)
    cq = J & 3
    X = surface_hidden(blob, p, cq - 1, True)
    v = dot_row16(X, blob.aux[2048:2304])
    sdf = np.zeros(4, np.float32); nab = np.zeros((4, 3), np.float32); h7 = np.zeros((4, 256), np.float32)
    for lane in range(64):
        pi = J[lane] >> 2
        if cq[lane] == 0:
            if G[lane] == 0:
                sv = v[lane] + blob.aux[2304]
                if R_bg > 0:
                    d_bg = R_bg - np.sqrt((p[lane] ** 2).sum())
                    sv = d_bg if d_bg < sv else sv
                sdf[pi] = sv
            for t in range(16):
                h7[pi, t * 16 + G[lane] * 4: t * 16 + G[lane] * 4 + 4] = X[t][lane]
        elif G[lane] == 0:
            nab[pi, cq[lane] - 1] = v[lane]
    return sdf, nab, h7


def emul_radiance(blob_np, view_tiles, pts16, view16, nabla16, h7_16):
    """[16,3] x3, h7 [16,256] -> rgb[16,3]"""
    blob = Blob(blob_np)
    p, v, n = pts16[J].astype(np.float32), view16[J].astype(np.float32), nabla16[J].astype(np.float32)
    X = [None] * 19
    for t in range(16):
        X[t] = np.stack([h7_16[J[l], t * 16 + G[l] * 4: t * 16 + G[l] * 4 + 4] for l in range(64)]).astype(np.float32)
    ne = 9 if view_tiles == 1 else 33
    ex = np.zeros((64, 16 * view_tiles), np.float32)
    ex[:, 0:3] = p
    if view_tiles == 1:
        ex[:, 3:6] = v
    else:
        ex[:, 3:6] = v
        for k in range(4):
            f = np.float32(1 << k)
            ex[:, 6 + 6 * k: 9 + 6 * k] = np.sin(v * f, dtype=np.float32)
            ex[:, 9 + 6 * k: 12 + 6 * k] = np.cos(v * f, dtype=np.float32)
    ex[:, ne - 3: ne] = n
    for t in range(view_tiles):
        X[16 + t] = np.stack([ex[l, 16 * t + 4 * G[l]: 16 * t + 4 * G[l] + 4] for l in range(64)])
    for L in range(5):
        run_layer(X, blob, blob.aux[L * 256:(L + 1) * 256], 16, view_tiles if L == 1 else 0, True,
                  "none" if L == 0 else "relu", False)
    rgb = np.zeros((16, 3), np.float32)
    for c in range(3):
        z = dot_row16(X, blob.aux[1280 + 256 * c: 1280 + 256 * (c + 1)]) + blob.aux[2048 + c]
        rgb[:, c] = (1.0 / (1.0 + np.exp(-z)))[:16]
    return rgb
