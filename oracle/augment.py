"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's rollout-time image augmentation
(core/carla_agent.py:545-577, rl/augmentations/augmentations.py, rl/augmentations/simclr.py:49-63).

Parity unpinned: TensorFlow cannot be imported here; the tf.image semantics are restated from their documented
behaviour (random_brightness: x + d; random_contrast: (x - mean_hw) * f + mean_hw per image and channel;
random_saturation / random_hue: RGB -> HSV, S * f clipped to [0, 1] / (H + d) mod 1, HSV -> RGB; nearest-neighbour
resize with half-pixel centres).  The random draws that decide WHAT is applied are explicit inputs (the plan); the
per-pixel random fields come from a Philox-4x32-10 counter RNG that the device kernels reproduce bit for bit, so the
GPU path is compared element by element.  Reference quirks kept on purpose: the "gaussian blur" kernel is a random
N(1, std) kernel that is not normalised (blur :185-197); gaussian noise is clipped to [0, 1] AFTER masking (only
positive noise is added, :139-150); cutout / coarse dropout use the FIRST image's mask for the whole stack (`[0]` after
the batched resize, :55-68, :80-91)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: (..., 4) uint32, key: (..., 2) uint32 -> (..., 4) uint32 (Random123 Philox-4x32-10)."""
    c = [np.asarray(counter[..., i], dtype=np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[..., 1], dtype=np.uint32).copy()
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[1] ^ k0
            n1 = (p1 & MASK32).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[3] ^ k1
            n3 = (p0 & MASK32).astype(np.uint32)
            c = [n0, n1, n2, n3]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def _block(seed, offset, idx):
    """First Philox block of the stream (seed, offset) at element index idx (same layout as csrc/sample.hip)."""
    idx = np.asarray(idx, dtype=np.uint64)
    ctr = np.stack([(idx & MASK32).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                    np.full(idx.shape, np.uint32(offset & 0xFFFFFFFF)), np.full(idx.shape, np.uint32((offset >> 32) & 0xFFFFFFFF))], axis=-1)
    key = np.stack([np.full(idx.shape, np.uint32(seed & 0xFFFFFFFF)), np.full(idx.shape, np.uint32((seed >> 32) & 0xFFFFFFFF))], axis=-1)
    return philox4x32_10(ctr, key)


def uniform(seed, offset, idx):
    """(0, 1) double from word 0 of the block."""
    return (_block(seed, offset, idx)[..., 0].astype(np.float64) + 0.5) / 4294967296.0


def normal(seed, offset, idx):
    """standard normal (Box-Muller on words 0, 1)."""
    b = _block(seed, offset, idx)
    u1 = (b[..., 0].astype(np.float64) + 0.5) / 4294967296.0
    u2 = (b[..., 1].astype(np.float64) + 0.5) / 4294967296.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586 * u2)


# stream ids (offset = plan offset * 8 + stream)
S_SP_SELECT, S_SP_NOISE, S_GN_SELECT, S_GN_NOISE, S_DROPOUT = 1, 2, 3, 4, 5


def rgb_to_hsv(x):
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    d = v - mn
    s = np.where(v > 0, d / np.where(v > 0, v, 1.0), 0.0)
    dn = np.where(d > 0, d, 1.0)
    h = np.where(v == r, (g - b) / dn, np.where(v == g, 2.0 + (b - r) / dn, 4.0 + (r - g) / dn))
    h = np.where(d > 0, h / 6.0, 0.0)
    h = h - np.floor(h)
    return np.stack([h, s, v], axis=-1)


def hsv_to_rgb(x):
    h, s, v = x[..., 0], x[..., 1], x[..., 2]
    h6 = h * 6.0
    k = lambda n: (n + h6) % 6.0
    f = lambda n: v - v * s * np.maximum(0.0, np.minimum(np.minimum(k(n), 4.0 - k(n)), 1.0))
    return np.stack([f(5.0), f(3.0), f(1.0)], axis=-1)


def color_jitter(x, brightness, contrast, saturation, hue):
    """simclr.color_jitter(original=True): brightness -> contrast -> saturation -> hue -> clip[0,1]; x: (T,H,W,3)."""
    x = x + brightness
    mean = x.mean(axis=(1, 2), keepdims=True)
    x = (x - mean) * contrast + mean
    hsv = rgb_to_hsv(x)
    hsv[..., 1] = np.clip(hsv[..., 1] * saturation, 0.0, 1.0)
    x = hsv_to_rgb(hsv)
    hsv = rgb_to_hsv(x)
    hsv[..., 0] = (hsv[..., 0] + hue) % 1.0
    x = hsv_to_rgb(hsv)
    return np.clip(x, 0.0, 1.0)


def blur(x, kernel):
    """depthwise conv, SAME zero padding, kernel (k,k,3), not normalised (as in the reference)."""
    k = kernel.shape[0]
    r = k // 2
    T, H, W, C = x.shape
    xp = np.zeros((T, H + 2 * r, W + 2 * r, C), x.dtype)
    xp[:, r:r + H, r:r + W] = x
    out = np.zeros_like(x)
    for ky in range(k):
        for kx in range(k):
            out += xp[:, ky:ky + H, kx:kx + W] * kernel[ky, kx]
    return out


def salt_and_pepper(x, amount, prob, seed, offset):
    T, H, W, _ = x.shape
    idx = np.arange(T * H * W, dtype=np.uint64).reshape(T, H, W)
    sel = (uniform(seed, offset * 8 + S_SP_SELECT, idx) < float(np.float32(amount) / np.float32(10.0))).astype(x.dtype)[..., None]
    noise = (uniform(seed, offset * 8 + S_SP_NOISE, idx) < float(np.float32(prob))).astype(x.dtype)[..., None]
    return x * (1.0 - sel) + noise * sel


def gaussian_noise(x, amount, std, seed, offset):
    T, H, W, C = x.shape
    pix = np.arange(T * H * W, dtype=np.uint64).reshape(T, H, W)
    sel = (uniform(seed, offset * 8 + S_GN_SELECT, pix) < float(np.float32(amount))).astype(np.float64)[..., None]
    el = np.arange(T * H * W * C, dtype=np.uint64).reshape(T, H, W, C)
    noise = normal(seed, offset * 8 + S_GN_NOISE, el) * float(np.float32(std))
    return x + np.clip(sel * noise, 0.0, 1.0)


def normalize(x, eps=np.finfo(np.float32).eps):
    x = x - x.min(axis=(1, 2, 3), keepdims=True)
    return x / (x.max(axis=(1, 2, 3), keepdims=True) + eps)


def _nearest(mask, H, W):
    s = mask.shape[0]
    iy = np.minimum(np.floor((np.arange(H) + 0.5) * s / H).astype(int), s - 1)
    ix = np.minimum(np.floor((np.arange(W) + 0.5) * s / W).astype(int), s - 1)
    return mask[iy][:, ix]


def cutout(x, size, cell):
    m = np.ones((size, size), x.dtype)
    m[cell // size, cell % size] = 0.0
    return x * _nearest(m, x.shape[1], x.shape[2])[None, :, :, None]


def coarse_dropout(x, size, amount, seed, offset):
    idx = np.arange(size * size, dtype=np.uint64).reshape(size, size)
    m = (uniform(seed, offset * 8 + S_DROPOUT, idx) < float(np.float32(1.0) - np.float32(amount))).astype(x.dtype)
    return x * _nearest(m, x.shape[1], x.shape[2])[None, :, :, None]


def augment(x, plan):
    """x: (T,H,W,3) float64/float32 in [0,1]; plan: dict as produced by rl.augmentations.draw_plan."""
    x = np.asarray(x, dtype=np.float64)
    seed, offset = plan['seed'], plan['offset']
    if plan.get('jitter'):
        x = color_jitter(x, plan['brightness'], plan['contrast'], plan['saturation'], plan['hue'])
    if plan.get('blur_size', 0):
        k = plan['blur_size']
        x = blur(x, np.asarray(plan['blur_kernel'], dtype=np.float64)[:3 * k * k].reshape(k, k, 3))
    if plan.get('salt_pepper'):
        x = salt_and_pepper(x, plan['sp_amount'], plan['sp_prob'], seed, offset)
    if plan.get('gauss_noise'):
        x = gaussian_noise(x, plan['gn_amount'], plan['gn_std'], seed, offset)
    if plan.get('normalize'):
        x = normalize(x)
    if plan.get('cutout_size', 0):
        x = cutout(x, plan['cutout_size'], plan['cutout_cell'])
    if plan.get('dropout_size', 0):
        x = coarse_dropout(x, plan['dropout_size'], plan['dropout_amount'], seed, offset)
    return x
