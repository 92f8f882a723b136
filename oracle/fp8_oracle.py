"""CPU statement of the FP8 forward path's numerics (TEST INFRASTRUCTURE ONLY - never imported by the product path).

BASELINE.json configs[4] names an "fp8 MFMA co-attention path". The reference has NO fp8 code (its reduced-precision
switch is apex fp16: /root/reference/train_tasks.py:168-171, vilbert/optimization.py FP16 wrappers), so there is no
reference behaviour to restate: PARITY UNPINNED against the reference by construction. What this file pins instead:

  * the number format: OCP 8-bit floating point E4M3 ("e4m3fn": 4 exponent bits, bias 7, 3 mantissa bits, no
    infinities, S.1111.111 = NaN, max finite 448, subnormals m/8 * 2^-6) - restated below from the OCP 8-bit
    Floating Point Specification (OFP8) rev 1.0, and checked bit-for-bit against PyTorch's independent CPU
    implementation ``torch.float8_e4m3fn`` over all 256 codes and on random data (tests/test_fp8_oracle.py);
  * the quantisation recipe of csrc/fp8.hip: one scale per ROW, scale = amax / 448, q = rne(x * (448 / amax));
  * the product: exact products of the quantised values (every e4m3 x e4m3 product is exact in fp32), summed - here
    in float64 - and multiplied by the two scales.

The fp8 model-level tolerance (how far an fp8 forward may drift from the fp32 forward) is a measured property stated
in tests/test_fp8_gpu.py, not a parity claim.
"""
import numpy as np

E4M3_MAX = 448.0


def e4m3_decode(codes):
    """uint8 codes -> float32 values (NaN for 0x7f / 0xff)."""
    c = np.asarray(codes, dtype=np.uint8).astype(np.int32)
    sign = np.where(c & 0x80, -1.0, 1.0)
    e = (c >> 3) & 0xF
    m = c & 0x7
    val = np.where(e == 0, (m / 8.0) * 2.0 ** -6, (1.0 + m / 8.0) * np.exp2(e.astype(np.float64) - 7.0))
    val = np.where((e == 15) & (m == 7), np.nan, val)
    return (sign * val).astype(np.float32)


def e4m3_encode(x):
    """float32 -> uint8 codes, round-to-nearest-even, saturating at +-448 (inputs of the recipe never exceed it)."""
    x = np.asarray(x, dtype=np.float32)
    a = np.abs(x).astype(np.float64)
    sign = (np.signbit(x)).astype(np.int32) << 7
    a = np.minimum(a, E4M3_MAX)
    # exponent of the binade (clamped to the subnormal binade -6), quantum = 2^(e - 3)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0))).astype(np.int32)
    e = np.clip(e, -6, 8)
    q = np.rint(a / np.exp2(e.astype(np.float64) - 3.0)).astype(np.int32)   # np.rint = ties to even; 0..16
    carry = q >= 16                                                       # rounded up into the next binade
    e = np.where(carry, e + 1, e)
    q = np.where(carry, 8, q)
    # q < 8 only in the subnormal binade (e == -6): code = q; otherwise (e + 7) << 3 | (q - 8)
    code = np.where(q < 8, q, ((e + 7) << 3) | (q - 8))
    code = np.minimum(code, 0x7E)                                         # saturate (448 = 0x7e)
    return (sign | code).astype(np.uint8)


def quantize_rows(x):
    """x [rows, K] float32 -> (codes uint8 [rows, K], scale float32 [rows]) exactly as quant_rows_kernel."""
    x = np.asarray(x, dtype=np.float32)
    amax = np.max(np.abs(x), axis=1).astype(np.float32)
    zero = ~(amax > 0)
    safe = np.where(zero, np.float32(1), amax).astype(np.float32)
    inv = np.where(zero, np.float32(1), np.float32(E4M3_MAX) / safe).astype(np.float32)      # fp32 division
    scale = np.where(zero, np.float32(1), safe / np.float32(E4M3_MAX)).astype(np.float32)
    y = (x * inv[:, None]).astype(np.float32)                                                # fp32 multiply
    return e4m3_encode(y), scale


def linear_fp8(x, w, bias=None):
    """Reference result of vb_linear_fwd_fp8 before the activation: float64 [M, N]."""
    qa, sa = quantize_rows(x)
    qw, sw = quantize_rows(w)
    prod = e4m3_decode(qa).astype(np.float64) @ e4m3_decode(qw).astype(np.float64).T
    y = prod * sa.astype(np.float64)[:, None] * sw.astype(np.float64)[None, :]
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float64)[None, :]
    return y


# ---------------------------------------------------------------------------------------------------------------------
# MX (OCP Microscaling Formats v1.0, "MXFP8 E4M3") statement of csrc/mx8.hip - round 4. Again not a reference feature
# (PARITY UNPINNED against the reference by construction); pinned here: the element format (e4m3fn, as above, against
# torch.float8_e4m3fn), the shared scale (E8M0: an 8-bit biased power of two, value 2^(byte - 127)) and the block of 32
# consecutive elements along the contraction dimension. The spec leaves the choice of the shared exponent to the
# producer; this path's rule is
#     e = the smallest integer with amax_block / 2^e <= 448   (nothing ever saturates; the spec's own example rule
#         floor(log2 amax) - 8 clips the top of the block instead),  byte = max(e + 127, 0)
# computed from the BITS of amax exactly as the kernels do (448 = 1.75 x 2^8), elements = e4m3_rne(x * 2^-e) - the scaling
# is exact, so there is exactly one rounding per element.
# ---------------------------------------------------------------------------------------------------------------------
MX_BLOCK = 32


def mx_scale_bytes(amax):
    """float32 block maxima (>= 0) -> uint8 E8M0 bytes."""
    u = np.asarray(amax, dtype=np.float32).view(np.uint32).astype(np.int64)
    b = (u >> 23) - 8 + ((u & 0x7FFFFF) > 0x600000)
    return np.clip(b, 0, 254).astype(np.uint8)


def mx_quantize(x):
    """x [rows, K] float32 (K % 32 == 0) -> (codes uint8 [rows, K], scale bytes uint8 [rows, K / 32])."""
    x = np.asarray(x, dtype=np.float32)
    rows, K = x.shape
    blocks = x.reshape(rows, K // MX_BLOCK, MX_BLOCK)
    byte = mx_scale_bytes(np.max(np.abs(blocks), axis=2))
    inv = np.ldexp(np.float32(1.0), 127 - byte.astype(np.int32)).astype(np.float32)       # 2^-e, exact
    y = (blocks * inv[:, :, None]).astype(np.float32)                                        # exact scaling
    return e4m3_encode(y).reshape(rows, K), byte


def mx_dequantize(codes, byte):
    """(codes [rows, K], scale bytes [rows, K / 32]) -> float64 values."""
    rows, K = codes.shape
    v = e4m3_decode(codes).astype(np.float64).reshape(rows, K // MX_BLOCK, MX_BLOCK)
    return (v * np.ldexp(1.0, byte.astype(np.int32) - 127)[:, :, None]).reshape(rows, K)


def mx_scale_words(byte, scale_rows=None):
    """Scale bytes [rows, K / 32] -> the kernels' layout: uint32 words [K / 128, scale_rows], word (kt, r) = bytes of row
    r's blocks 4 kt .. 4 kt + 3 (little endian: byte b of the word = block 4 kt + b); rows beyond `rows` are zero here
    (the kernels leave them unwritten)."""
    rows, nb = byte.shape
    assert nb % 4 == 0
    scale_rows = rows if scale_rows is None else scale_rows
    w = byte.reshape(rows, nb // 4, 4).astype(np.uint32)
    words = w[:, :, 0] | (w[:, :, 1] << 8) | (w[:, :, 2] << 16) | (w[:, :, 3] << 24)
    out = np.zeros((nb // 4, scale_rows), dtype=np.uint32)
    out[:, :rows] = words.T
    return out


def mx_words_to_bytes(words, rows):
    """Inverse of mx_scale_words for the first `rows` rows."""
    w = np.asarray(words, dtype=np.uint32)[:, :rows].T            # [rows, K / 128]
    b = np.stack([(w >> (8 * i)) & 0xFF for i in range(4)], axis=2)
    return b.reshape(rows, -1).astype(np.uint8)


def linear_mx(x, w, bias=None):
    """Reference result of vb_linear_fwd_mx before the activation: float64 [M, N] (exact sums of the dequantised
    operands; the kernel accumulates in fp32 inside the scaled MFMA)."""
    qa, sa = mx_quantize(x)
    qw, sw = mx_quantize(w)
    y = mx_dequantize(qa, sa) @ mx_dequantize(qw, sw).T
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float64)[None, :]
    return y


MX_GELU_COEF = (0.398325773, -0.0648922966, 0.00876406186, -0.000774126121, 3.89366778e-05, -8.3218473e-07)


def mx_gelu(x):
    """GELU of the MX GEMM epilogues (csrc/mx8.h mx_gelu): x * (0.5 + t Q(t^2)), t = clamp(x, -3.5, 3.5), Q of degree 5 -
    within 4e-4 |x| + 6e-4 of the reference's erf form (vilbert.py:111-117), float64 here."""
    x = np.asarray(x, dtype=np.float64)
    t = np.clip(x, -3.5, 3.5)
    u = t * t
    q = np.zeros_like(u)
    for c in reversed(MX_GELU_COEF):
        q = q * u + c
    return x * (0.5 + t * q)
