"""Second, independent restatement (numpy) of the dense stages of the reference, used only to
cross-check the C oracle (two independent restatements reduce the risk of a shared
misreading). Written from the reference shaders / Vulkan rules, deliberately in a different
formulation than oracle/sift_oracle.c:

  * the sampler-interpolated blur is evaluated as *bilinear texture fetches* at fractional offsets
    (GaussianBlurInterpolated.comp:32-44), not as expanded direct taps
  * the 2x up-sampling uses the closed form 0.25/0.75 weights, the down-sampling the odd texels
  * the matcher ranks a full distance matrix
"""
import numpy as np

f32 = np.float32


def gaussian_kernel(sigma):
    """sift_detector.c:92-117"""
    sigma = f32(sigma)
    k = int(np.ceil(sigma * f32(4)) + 1)
    k = min(k, 20)
    w = np.ones(k, dtype=f32)
    for i in range(1, k):
        w[i] = f32(np.exp(-0.5 * float(f32(i) ** 2) / float(sigma ** 2)))
    total = f32(w[0])
    for i in range(1, k):
        total = f32(total + f32(2) * w[i])
    return (w / total).astype(f32)


def scale_sigmas(S=3, seed=1.6, in_blur=0.5, ups=True):
    """sift_detector.c:76-89"""
    seed, in_blur = f32(seed), f32(in_blur)
    init = in_blur * f32(2) if ups else in_blur
    out = [np.sqrt(seed * seed - init * init, dtype=f32)]
    k = f32(2.0) ** (f32(1) / f32(S))
    for i in range(1, S + 3):
        prev = f32(k ** f32(i - 1)) * seed
        tot = f32(prev * k)
        out.append(np.sqrt(tot * tot - prev * prev, dtype=f32))
    return out


def mirror(idx, n):
    """VK_SAMPLER_ADDRESS_MODE_MIRRORED_REPEAT on integer texel indices"""
    idx = np.mod(idx, 2 * n)
    return np.where(idx < n, idx, 2 * n - 1 - idx)


def _fetch_bilinear_1d(img, axis, offset):
    """texture fetch at texel-centre + offset along `axis` with LINEAR filtering and mirrored repeat"""
    n = img.shape[axis]
    base = int(np.floor(offset))
    frac = f32(offset - base)
    i = np.arange(n)
    a = np.take(img, mirror(i + base, n), axis=axis)
    b = np.take(img, mirror(i + base + 1, n), axis=axis)
    return (a * (f32(1) - frac) + b * frac).astype(f32)


def blur_pass(img, w, axis, interpolated):
    K = len(w)
    out = (img * w[0]).astype(f32)
    if not interpolated:
        for i in range(1, K):
            out = out + (_fetch_bilinear_1d(img, axis, i) + _fetch_bilinear_1d(img, axis, -i)) * w[i]
    else:
        d = 1
        while d + 1 < K:  # host pairing loop, sift_detector.c:130
            c = f32(w[d] + w[d + 1])
            off = f32((f32(d) * w[d] + f32(d + 1) * w[d + 1]) / c)
            out = out + (_fetch_bilinear_1d(img, axis, float(off)) + _fetch_bilinear_1d(img, axis, -float(off))) * c
            d += 2
    return out.astype(f32)


def blur(img, sigma, interpolated):
    w = gaussian_kernel(sigma)
    return blur_pass(blur_pass(img, w, 1, interpolated), w, 0, interpolated)


def upsample2x(u8):
    """vkCmdBlitImage LINEAR, exact 2x: dst[2k] = .25 s[k-1] + .75 s[k]; dst[2k+1] = .75 s[k] + .25 s[k+1]; clamp to edge"""
    s = u8.astype(f32) / f32(255)

    def up(a, axis):
        n = a.shape[axis]
        i = np.arange(n)
        prev = np.take(a, np.clip(i - 1, 0, n - 1), axis=axis)
        nxt = np.take(a, np.clip(i + 1, 0, n - 1), axis=axis)
        even = f32(0.25) * prev + f32(0.75) * a
        odd = f32(0.75) * a + f32(0.25) * nxt
        out = np.stack([even, odd], axis=axis + 1)
        shape = list(a.shape)
        shape[axis] *= 2
        return out.reshape(shape).astype(f32)

    return up(up(s, 1), 0)


def downsample_nearest(img, dw, dh):
    """vkCmdBlitImage NEAREST 2:1 -> odd texels"""
    return img[1::2, 1::2][:dh, :dw].copy()


def build_octave0(u8, S=3, interpolated=True, ups=True):
    sig = scale_sigmas(S, ups=ups)
    g = [blur(upsample2x(u8) if ups else u8.astype(f32) / f32(255), sig[0], interpolated)]
    for s in range(1, S + 3):
        g.append(blur(g[-1], sig[s], interpolated))
    d = [g[s + 1] - g[s] for s in range(S + 2)]
    return g, d


def match_2nn(a, b):
    """Get2NearestNeighbors.comp: float sqrt distances, strict '<' scan in index order, b0/b1 init."""
    a = a.astype(np.int64)
    b = b.astype(np.int64)
    d2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2 * a @ b.T
    d = np.sqrt(d2.astype(f32)).astype(f32)
    out = []
    for i in range(len(a)):
        row = d[i]
        if row[0] < row[1]:
            bi, si = 0, 1
        else:
            bi, si = 1, 0
        bd, sd = row[bi], row[si]
        for j in range(2, len(b)):
            if row[j] < bd:
                sd, si = bd, bi
                bd, bi = row[j], j
            elif row[j] < sd:
                sd, si = row[j], j
        out.append((i, bi, si, bd, sd))
    return out


# ---------------------------------------------------------------------------------------------------------------
# whole pyramid (all octaves), so that tests/np_features.py can run on planes the oracle never touched
# ---------------------------------------------------------------------------------------------------------------
def octave_resolutions(w, h, ups=True, nb_octaves=0):
    """sift_memory.c:15-38: n = (uint)(log2f(min(w, h)) - 4 + ups); w_o = (uint)((1 / (2^o * sf)) * w), sf = 0.5 if ups"""
    n = int(f32(np.log2(f32(min(w, h)))) - f32(4) + f32(1 if ups else 0))
    if nb_octaves > 0:
        n = min(n, nb_octaves)
    sf = f32(0.5) if ups else f32(1)
    out = []
    for o in range(n):
        k = f32(1) / f32(f32(2 ** o) * sf)
        out.append((int(f32(k * f32(w))), int(f32(k * f32(h)))))
    return out


def blit_nearest(src, dw, dh):
    """vkCmdBlitImage NEAREST: dst(x, y) = src(floor((x + .5) * Ws / Wd), floor((y + .5) * Hs / Hd)), in exact rationals"""
    sh, sw = src.shape
    xs = ((2 * np.arange(dw) + 1) * sw) // (2 * dw)
    ys = ((2 * np.arange(dh) + 1) * sh) // (2 * dh)
    return src[np.ix_(ys, xs)].copy()


def build_pyramid(u8, S=3, seed=1.6, in_blur=0.5, ups=True, interpolated=True, nb_octaves=0):
    """[(gauss (S+3,H,W), dog (S+2,H,W))] per octave: sift_detector.c:893-1037 schedule"""
    h, w = u8.shape
    res = octave_resolutions(w, h, ups, nb_octaves)
    sig = scale_sigmas(S, seed, in_blur, ups)
    out = []
    base = upsample2x(u8) if ups else (u8.astype(f32) / f32(255))
    for o, (ow, oh) in enumerate(res):
        if o == 0:
            assert base.shape == (oh, ow)
            g = [blur(base, sig[0], interpolated)]
        else:
            g = [blit_nearest(out[-1][0][S], ow, oh)]      # Gaussian layer S of the previous octave (blur 2 * seed)
        for s in range(1, S + 3):
            g.append(blur(g[-1], sig[s], interpolated))
        g = np.stack(g)
        out.append((g, (g[1:] - g[:-1]).astype(f32)))
    return out
