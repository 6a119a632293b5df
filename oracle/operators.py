"""Oracle: linear forward operators (test infrastructure, see oracle/__init__.py).

The reference writes every operator in SVD form, H = U diag(s) V^T, and applies
it as `U(s * Vt(x)[:, :len(s)])` (obs_functions/Hfuncs.py:65-71), H^T as
`V(add_zeros(s * Ut(y)))` (:73-79) and H^+ with 1/s on the non-zero singular
values (:81-90).  This file restates WHAT those compositions compute for the
three operators on the hot path, in plain fp32 torch-CPU ops:

  InpaintRef      Hfuncs.py:119-154  gather of the HWC-interleaved image at the
                                     ascending kept indices; H^T = H^+ = scatter
  BlockMeanRef    Hfuncs.py:180-234  r x r block mean, output flattened CHW;
                                     H^T = nearest-upsample / r^2, H^+ = nearest-upsample
  SpectralBlurRef Hfuncs.py:448-523  Y_c = U1 (D_c o (V1^T X_c V2)) U2^T with the
                                     per-channel multiplier map D_c the reference's
                                     tiled-singulars / interleaved-Vt layout produces

All take [B,C,H,W] (or anything reshapeable to it) and return [B, M] for H and
[B, C*H*W] for H^T / H^+, like the reference.
"""
import math
import torch


class InpaintRef:
    """Hfuncs.py:119-154.  `missing` are indices into the HWC-flattened image.

    Vt() permutes CHW -> HWC and moves kept entries first (:134-139); the
    singular values are all 1 (:123); U is the identity (:141-145).  Hence
    H(x)[k] = x_hwc[kept[k]] with kept ascending (:125).
    """

    def __init__(self, channels, img_dim, missing):
        self.channels, self.img_dim = channels, img_dim
        n = channels * img_dim * img_dim
        keep = torch.ones(n, dtype=torch.bool)
        keep[missing.long().cpu()] = False
        self.kept = torch.nonzero(keep).squeeze(1)      # ascending, int64
        self.missing = missing.long().cpu()
        self.M = int(self.kept.numel())

    def _hwc(self, x):
        B = x.shape[0]
        return x.reshape(B, self.channels, -1).permute(0, 2, 1).reshape(B, -1)

    def H(self, x):
        return self._hwc(x)[:, self.kept]

    def Ht(self, y):
        B = y.shape[0]
        hwc = torch.zeros(B, self.channels * self.img_dim ** 2, dtype=y.dtype)
        hwc[:, self.kept] = y.reshape(B, -1)
        return hwc.reshape(B, -1, self.channels).permute(0, 2, 1).reshape(B, -1)

    H_pinv = Ht          # all singular values are 1


class BlockMeanRef:
    """Hfuncs.py:180-234 (SuperResolution).

    SVD of the 1 x r^2 row [1/r^2]*r^2 (:187-188): s = 1/r, V[:,0] = +-1/r, U = +-1.
    Vt() extracts r x r patches and keeps the first coefficient of each (:206-219),
    so H(x) = U00 * (1/r) * sum_i(V[i,0] * patch_i) = block mean, laid out
    [c, i, j] (CHW).  H^T spreads y/r^2 over the block, H^+ spreads y.
    """

    def __init__(self, channels, img_dim, ratio):
        assert img_dim % ratio == 0
        self.channels, self.img_dim, self.ratio = channels, img_dim, ratio
        self.y_dim = img_dim // ratio
        self.M = channels * self.y_dim ** 2

    def H(self, x):
        B, r, d = x.shape[0], self.ratio, self.y_dim
        blocks = x.reshape(B, self.channels, d, r, d, r)
        return blocks.sum(dim=(3, 5)).mul(1.0 / (r * r)).reshape(B, -1)

    def _up(self, y, scale):
        B, r, d = y.shape[0], self.ratio, self.y_dim
        img = y.reshape(B, self.channels, d, 1, d, 1).expand(B, self.channels, d, r, d, r)
        return (img * scale).reshape(B, -1)

    def Ht(self, y):
        return self._up(y, 1.0 / (self.ratio ** 2))

    def H_pinv(self, y):
        return self._up(y, 1.0)


def band_matrix(kernel, img_dim):
    """Hfuncs.py:459-471: 1-D convolution matrix.  The reference's loop runs
    j in [i - k//2, i + k//2) -- the upper bound is exclusive, so a 9-tap kernel
    contributes only its first 8 taps."""
    k = kernel.shape[0]
    Hs = torch.zeros(img_dim, img_dim)
    for i in range(img_dim):
        for j in range(i - k // 2, i + k // 2):
            if 0 <= j < img_dim:
                Hs[i, j] = kernel[j - i + k // 2]
    return Hs


def gaussian_taps(sigma, half=4):
    """main_sampling.py:327-335: exp(-0.5 (x/sigma)^2) for x in -4..4, normalised."""
    taps = torch.tensor([math.exp(-0.5 * (x / sigma) ** 2) for x in range(-half, half + 1)],
                        dtype=torch.float32)
    return taps / taps.sum()


class SpectralBlurRef:
    """Hfuncs.py:448-523 (Deblurring2D), carried as DATA.

    Fields: U1,U2,V1,V2 [d,d] and D [C,d,d].  kernel1 acts along rows (height,
    left-multiply), kernel2 along width (right-multiply) (:486-487).

    `singulars()` tiles the sorted products 3x (:519-520) while Vt() interleaves
    channels (:493-499), so spectral position perm[k] of channel c is multiplied
    by s_sorted[(3k+c) mod d^2]; `from_kernels` builds that map.
    """

    def __init__(self, U1, U2, V1, V2, D):
        self.U1, self.U2, self.V1, self.V2, self.D = U1, U2, V1, V2, D
        self.channels, self.img_dim = D.shape[0], D.shape[1]
        self.M = D.numel()

    @classmethod
    def from_kernels(cls, kernel1, kernel2, channels, img_dim, zero=3e-2, stable=False):
        H1, H2 = band_matrix(kernel1, img_dim), band_matrix(kernel2, img_dim)
        U1, s1, V1 = torch.svd(H1, some=False)
        U2, s2, V2 = torch.svd(H2, some=False)
        s1 = torch.where(s1 < zero, torch.zeros_like(s1), s1)     # :475-477
        s2 = torch.where(s2 < zero, torch.zeros_like(s2), s2)
        prod = torch.matmul(s1.reshape(img_dim, 1), s2.reshape(1, img_dim)).reshape(-1)
        s_sorted, perm = prod.sort(descending=True, stable=stable)  # :481 (unstable there)
        return cls(U1, U2, V1, V2, cls.multiplier_map(s_sorted, perm, channels, img_dim))

    @staticmethod
    def multiplier_map(s_sorted, perm, channels, img_dim):
        hw = img_dim * img_dim
        k = torch.arange(hw)
        D = torch.zeros(channels, hw, dtype=s_sorted.dtype)
        for c in range(channels):
            D[c, perm] = s_sorted[(channels * k + c) % hw]
        return D.reshape(channels, img_dim, img_dim)

    def _img(self, v):
        return v.reshape(v.shape[0], self.channels, self.img_dim, self.img_dim)

    def _sandwich(self, L, X, R, Dmap, Lo, Ro):
        spec = torch.matmul(torch.matmul(L.t(), X), R) * Dmap
        return torch.matmul(torch.matmul(Lo, spec), Ro.t())

    def H(self, x):
        X = self._img(x)
        return self._sandwich(self.V1, X, self.V2, self.D, self.U1, self.U2).reshape(X.shape[0], -1)

    def Ht(self, y):
        Y = self._img(y)
        return self._sandwich(self.U1, Y, self.U2, self.D, self.V1, self.V2).reshape(Y.shape[0], -1)

    def H_pinv(self, y):
        Y = self._img(y)
        Dp = torch.where(self.D != 0, 1.0 / self.D, torch.zeros_like(self.D))
        return self._sandwich(self.U1, Y, self.U2, Dp, self.V1, self.V2).reshape(Y.shape[0], -1)


class ColorRef:
    """Hfuncs.py:655-695 (Colorization) in the reference's rounding order.  With (u, s, V) = svd([[0.3333, 0.3334,
    0.3333]]) (:660-661) and v = V[:, 0]:  Vt (:673-680) is a [3 x 3] @ [3 x 1] matmul per pixel, whose K = 3 products
    torch's CPU kernel rounds one by one and sums left to right; H (:65-71) then multiplies by the singular value and
    by U[0,0]:  H x = u * (s * ((v0 x0 + v1 x1) + v2 x2)).  H^T y = v_c * (s * (u * y)) (:73-78 with the zero padding
    of :691-695 contributing exact zeros), H^+ y = v_c * ((u * y) * (1 / s)) (:80-90).  Reproduces the reference's whole
    `hmc()` run bit for bit (G15 color)."""

    def __init__(self, img_dim):
        self.channels, self.img_dim, self.M = 3, img_dim, img_dim * img_dim
        U, s, V = torch.svd(torch.Tensor([[0.3333, 0.3334, 0.3333]]), some=False)
        self.s, self.u, self.v = s[0], U[0, 0], V[:, 0].clone()
        self.w = (U[0, 0] * s[0]) * V[:, 0]                       # the collapsed weights, for reference only

    def H(self, x):
        X = x.reshape(x.shape[0], 3, -1)
        spec = (self.v[0] * X[:, 0] + self.v[1] * X[:, 1]) + self.v[2] * X[:, 2]
        return self.u * (self.s * spec)

    def Ht(self, y):
        B = y.shape[0]
        t = self.s * (self.u * y.reshape(B, 1, -1))
        return (self.v.view(1, 3, 1) * t).reshape(B, -1)

    def H_pinv(self, y):
        B = y.shape[0]
        t = (self.u * y.reshape(B, 1, -1)) * (1 / self.s)             # :86-88 multiplies by the reciprocal
        return (self.v.view(1, 3, 1) * t).reshape(B, -1)


class WalshHadamardRef:
    """Hfuncs.py:611-651 (WalshHadamardCS): orthonormal FWHT per channel image (butterflies h = 1, 2, 4, ... on the
    row-major flattened image, then / img_dim, :613-623); rows perm[:d^2/ratio] are kept, interleaved (k, c)."""

    def __init__(self, channels, img_dim, ratio, perm):
        self.channels, self.img_dim, self.ratio, self.perm = channels, img_dim, ratio, perm.long()
        self.M = channels * img_dim ** 2 // ratio

    def fwht(self, vec):
        B, n = vec.shape[0], self.img_dim ** 2
        a = vec.reshape(B, self.channels, n)
        h = 1
        while h < n:
            a = a.reshape(B, self.channels, -1, 2 * h)
            lo, hi = a[..., :h], a[..., h:]
            a = torch.cat([lo + hi, lo - hi], dim=-1)
            h *= 2
        return a.reshape(B, self.channels, n) / self.img_dim

    def H(self, x):
        B = x.shape[0]
        return self.fwht(x)[:, :, self.perm].permute(0, 2, 1).reshape(B, -1)[:, :self.M]

    def Ht(self, y):
        B, n = y.shape[0], self.img_dim ** 2
        full = torch.zeros(B, self.channels * n, dtype=y.dtype)
        full[:, :self.M] = y.reshape(B, -1)
        spec = torch.zeros(B, self.channels, n, dtype=y.dtype)
        spec[:, :, self.perm] = full.reshape(B, n, self.channels).permute(0, 2, 1)
        return self.fwht(spec).reshape(B, -1)

    H_pinv = Ht


class SeparableStridedRef:
    """Hfuncs.py:527-607 (SRConv) in the reference's stage order.  With U diag(s) V^T the SVD of the strided 1-D kernel
    matrix (reflective padding :547-551; s < 3e-2 zeroed :557-558), V1 = V[:, :sd], S = s s^T (:560):
        H(X)   = U (S o (V1^T X V1)) U^T     (H :65-71 = U(singulars * Vt(x)); Vt :576-583 multiplies from the left first,
        H^T(Y) = V1 (S o (U^T Y U)) V1^T      then from the right; U :585-592, Ut :594-596, V :566-574 likewise)
        H^+(Y) = V1 (S+ o (U^T Y U)) V1^T    (S+ = 1 / S where S != 0, :82-88)
    The top-left sd x sd block the reference's permutation selects (:563-564) is the product with V1; its entries are
    the same dot products, so slicing before or after the matmul changes nothing.  On this torch build the three maps
    reproduce G10 bit for bit."""

    def __init__(self, kernel, channels, img_dim, stride, svd=None):
        self.channels, self.img_dim, self.small_dim = channels, img_dim, img_dim // stride
        k, sd = kernel.shape[0], self.small_dim
        Hs = torch.zeros(sd, img_dim)
        for i in range(stride // 2, img_dim + stride // 2, stride):
            for j in range(i - k // 2, i + k // 2):
                je = j
                if je < 0:
                    je = -je - 1
                if je >= img_dim:
                    je = (img_dim - 1) - (je - img_dim)
                Hs[i // stride, je] += kernel[j - i + k // 2]
        U, s, V = svd if svd is not None else torch.svd(Hs, some=False)       # svd: a reference instance's exported factors
        s = s.clone()
        s[s < 3e-2] = 0
        self.U, self.V = U, V                                       # V: full [d, d], as the reference multiplies by it
        self.S = torch.matmul(s.reshape(sd, 1), s.reshape(1, sd))
        self.Sinv = self.S.clone()
        self.Sinv[self.S != 0] = 1 / self.S[self.S != 0]
        self.M = channels * sd * sd

    def _vt(self, X):
        sd = self.small_dim
        return torch.matmul(torch.matmul(self.V.t(), X), self.V)[..., :sd, :sd]

    def _v(self, W):
        sd, d = self.small_dim, self.img_dim
        full = torch.zeros(W.shape[:-2] + (d, d), dtype=W.dtype)
        full[..., :sd, :sd] = W
        return torch.matmul(torch.matmul(self.V, full), self.V.t())

    def H(self, x):
        X = x.reshape(x.shape[0], self.channels, self.img_dim, self.img_dim)
        Z = self.S * self._vt(X)
        return torch.matmul(torch.matmul(self.U, Z), self.U.t()).reshape(x.shape[0], -1)

    def _adjoint(self, y, mul):
        Y = y.reshape(y.shape[0], self.channels, self.small_dim, self.small_dim)
        W = mul * torch.matmul(torch.matmul(self.U.t(), Y), self.U)
        return self._v(W).reshape(y.shape[0], -1)

    def Ht(self, y):
        return self._adjoint(y, self.S)

    def H_pinv(self, y):
        return self._adjoint(y, self.Sinv)


def random_inpaint_missing(img_dim, frac=0.92, generator=None):
    """main_sampling.py:302-305: whole RGB triples at randperm(H*W)[:0.92 H*W]."""
    hw = img_dim * img_dim
    r = 3 * torch.randperm(hw, generator=generator)[: int(hw * frac)].long()
    return torch.cat([r, r + 1, r + 2], dim=0)
