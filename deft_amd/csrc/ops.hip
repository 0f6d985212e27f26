// Non-GEMM kernels of the DEFT hot path for gfx950: layout adapters, pooling,
// transposed-conv upsample + skip add, peak extraction / top-K, sparse heads,
// box decode, sparse embedding head, dual-softmax affinity tail, batched LSTM step.
// All HBM-bound or tiny: coalesced float4 NHWC traffic, LDS staging, wave shuffles.
#include <cstdarg>

#include "common.h"

static thread_local char g_err[512] = "";

void deft_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* deft_last_error(void) { return g_err; }
extern "C" int deft_version(void) { return DEFT_ABI_VERSION; }
extern "C" int deft_pieces(void) { return DEFT_NP; }

// ---------------------------------------------------------------------------
// layout adapters
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int N, int C, int HW, int ldy) {
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (long long)N * HW) return;
    const int n = (int)(pix / HW);
    const int hw = (int)(pix - (long long)n * HW);
    const float* xp = x + (size_t)n * C * HW + hw;
    float* yp = y + (size_t)pix * ldy;
    for (int c = 0; c < ldy; ++c) yp[c] = c < C ? xp[(size_t)c * HW] : 0.f;
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int N, int C, int HW, int ldx) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * C * HW) return;
    const int hw = (int)(i % HW);
    const long long t = i / HW;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    y[i] = x[((size_t)n * HW + hw) * ldx + c];
}

extern "C" int deft_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int ldy, void* stream) {
    DEFT_CHECK(x && y && ldy >= C, -1, "deft_nchw_to_nhwc: bad arguments");
    const long long tot = (long long)N * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(deft_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, H * W, ldy);
    DEFT_CHECK_LAUNCH("nchw_to_nhwc");
    return 0;
}
extern "C" int deft_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int ldx, void* stream) {
    DEFT_CHECK(x && y && ldx >= C, -1, "deft_nhwc_to_nchw: bad arguments");
    const long long tot = (long long)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(deft_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, H * W, ldx);
    DEFT_CHECK_LAUNCH("nhwc_to_nchw");
    return 0;
}

// ---------------------------------------------------------------------------
// Pre-processing on the device (detector.py:346-422): cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) of the uint8 HWC
// frame into the network input grid + ((v / 255 - mean) / std) + HWC -> NHWC(ld), ONE pass: 3 B/pixel in instead of the
// 12 B/pixel fp32 NCHW image (and no nchw_to_nhwc pass).  cv2's fixed-point arithmetic restated (cv2 is absent here: parity
// unpinned): source coordinates in 1/1024 px (AB_BITS = 10) from rounded per-column / per-row terms + round_delta 16, cut to 1/32 px
// (INTER_BITS = 5); bilinear weights = products of the 1/32 fractions scaled to 2^15 (exact integers, sum 2^15); result
// (sum + 2^14) >> 15.  The normalisation is a 3 x 256 table computed by the host in float64 like the reference's numpy expression.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ src, int sh, int sw, const double* __restrict__ minv,
                                                            const float* __restrict__ lut, float* __restrict__ y, int N, int H, int W, int ldy) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H * W) return;
    const int x = (int)(i % W);
    const long long t = i / W;
    const int yy = (int)(t % H), n = (int)(t / H);
    const double* m = minv + n * 6;
    const int X0 = deft_rint(m[0] * x * 1024.0) + deft_rint((m[1] * yy + m[2]) * 1024.0) + 16;
    const int Y0 = deft_rint(m[3] * x * 1024.0) + deft_rint((m[4] * yy + m[5]) * 1024.0) + 16;
    const int X = X0 >> 5, Y = Y0 >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    const unsigned char* base = src + (size_t)n * sh * sw * 3;
    const bool x0 = (unsigned)sx < (unsigned)sw, x1 = (unsigned)(sx + 1) < (unsigned)sw, y0 = (unsigned)sy < (unsigned)sh, y1 = (unsigned)(sy + 1) < (unsigned)sh;
    float* yp = y + (size_t)i * ldy;
    float o[3];
    if (sx >= 0 && sy >= 0 && sx + 2 < sw && sy + 1 < sh) {
        // interior: the two pixels of a source row are 6 consecutive bytes -- ONE (unaligned) 8-byte load per row instead of 6 byte loads
        unsigned long long r0, r1;
        __builtin_memcpy(&r0, base + ((size_t)sy * sw + sx) * 3, 8);
        __builtin_memcpy(&r1, base + ((size_t)(sy + 1) * sw + sx) * 3, 8);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p00 = (int)((r0 >> (8 * c)) & 255), p01 = (int)((r0 >> (8 * c + 24)) & 255);
            const int p10 = (int)((r1 >> (8 * c)) & 255), p11 = (int)((r1 >> (8 * c + 24)) & 255);
            const int v = (w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11 + (1 << 14)) >> 15;
            o[c] = lut[c * 256 + v];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p00 = (x0 && y0) ? base[((size_t)sy * sw + sx) * 3 + c] : 0;
            const int p01 = (x1 && y0) ? base[((size_t)sy * sw + sx + 1) * 3 + c] : 0;
            const int p10 = (x0 && y1) ? base[((size_t)(sy + 1) * sw + sx) * 3 + c] : 0;
            const int p11 = (x1 && y1) ? base[((size_t)(sy + 1) * sw + sx + 1) * 3 + c] : 0;
            const int v = (w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11 + (1 << 14)) >> 15;
            o[c] = lut[c * 256 + v];
        }
    }
    if (ldy == 4) {
        *(float4*)yp = make_float4(o[0], o[1], o[2], 0.f);            // (y is a plan buffer: 16-byte aligned, ld = 4)
    } else {
        yp[0] = o[0]; yp[1] = o[1]; yp[2] = o[2];
        for (int c = 3; c < ldy; ++c) yp[c] = 0.f;
    }
}

extern "C" int deft_preprocess_u8(const unsigned char* src, int N, int sh, int sw, const double* minv, const float* lut, float* y, int H, int W, int ldy, void* stream) {
    DEFT_CHECK(src && minv && lut && y && N > 0 && sh > 0 && sw > 0 && H > 0 && W > 0 && ldy >= 3, -1, "deft_preprocess_u8: bad arguments");
    DEFT_CHECK((long long)sh * sw * 3 < (1ll << 31), -2, "deft_preprocess_u8: source frame too large");
    const long long tot = (long long)N * H * W;
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(deft_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, src, sh, sw, minv, lut, y, N, H, W, ldy);
    DEFT_CHECK_LAUNCH("preprocess_u8");
    return 0;
}

// ---------------------------------------------------------------------------
// MaxPool2d(2,2)   (dla.py:266-267)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int N, int H, int W, int C4, int ldx, int ldy) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int OH = H >> 1, OW = W >> 1;
    if (i >= (long long)N * OH * OW * C4) return;
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    const float* p = x + ((size_t)(n * H + 2 * oy) * W + 2 * ox) * ldx + 4 * c4;
    const float4 a = *(const float4*)p, b = *(const float4*)(p + ldx);
    const float4 c = *(const float4*)(p + (size_t)W * ldx), d = *(const float4*)(p + (size_t)W * ldx + ldx);
    float4 r;
    r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
    r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
    r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    *(float4*)(y + ((size_t)(n * OH + oy) * OW + ox) * ldy + 4 * c4) = r;
}

extern "C" int deft_maxpool2x2(const float* x, float* y, int N, int H, int W, int C, int ldx, int ldy, void* stream) {
    DEFT_CHECK(x && y && (C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && (H & 1) == 0 && (W & 1) == 0, -1,
               "deft_maxpool2x2: need C,ld %% 4 == 0 and even H,W (got C=%d H=%d W=%d)", C, H, W);
    const long long tot = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(deft_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C / 4, ldx, ldy);
    DEFT_CHECK_LAUNCH("maxpool2x2");
    return 0;
}

// ---------------------------------------------------------------------------
// depthwise ConvTranspose2d(k=2f, stride f, pad f/2) + skip   (dla.py:677-699)
// out[oy] gathers in[iy]*w[ky] with ky = oy + f/2 - iy*f in [0,2f): two rows, two cols.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ x, const float* __restrict__ wup,
                                                           const float* __restrict__ skip, float* __restrict__ y, deft_piece_t* __restrict__ y3,
                                                           int N, int H, int W, int C4, int f, int ldx, int lds, int ldy, int ldy3) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int OH = H * f, OW = W * f;
    if (i >= (long long)N * OH * OW * C4) return;
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    const int k2 = 2 * f;
    const int py = oy + f / 2, px = ox + f / 2;
    const int iy1 = py / f, ix1 = px / f;            // tap ky1 = py - iy1*f in [0,f)
    const int ky1 = py - iy1 * f, kx1 = px - ix1 * f;
    const size_t o = ((size_t)(n * OH + oy) * OW + ox);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int iy = iy1 - 1 + a, ky = ky1 + f - a * f;   // a=0: (iy1-1, ky1+f); a=1: (iy1, ky1)
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ix = ix1 - 1 + b, kx = kx1 + f - b * f;
            if (ix < 0 || ix >= W) continue;
            const float4 v = *(const float4*)(x + ((size_t)(n * H + iy) * W + ix) * ldx + 4 * c4);
            const float4 w = *(const float4*)(wup + (size_t)(ky * k2 + kx) * (4 * C4) + 4 * c4);   // [tap][C]
            acc[0] += v.x * w.x;
            acc[1] += v.y * w.y;
            acc[2] += v.z * w.z;
            acc[3] += v.w * w.w;
        }
    }
    const float4 s = *(const float4*)(skip + o * lds + 4 * c4);
    const f32x4 r = {acc[0] + s.x, acc[1] + s.y, acc[2] + s.z, acc[3] + s.w};
    *(f32x4*)(y + o * ldy + 4 * c4) = r;
    if (y3 != nullptr) {                       // the operand pieces for a following pre-split conv (DeftGemmDesc.x3 layout)
        pcx4 pc[DEFT_NP];
        deft_split(r, pc, DEFT_ASCALE);
        const int c = 4 * c4;
        deft_piece_t* yp = y3 + o * ldy3 * DEFT_NP + (c >> 5) * (32 * DEFT_NP) + (c & 31);
#pragma unroll
        for (int q = 0; q < DEFT_NP; ++q) *(pcx4*)(yp + 32 * q) = pc[q];
    }
}

extern "C" int deft_upsample_add(const float* x, const float* wup, const float* skip, float* y,
                                 int N, int H, int W, int C, int f, int ldx, int lds, int ldy, void* y3, int ldy3, void* stream) {
    DEFT_CHECK(y3 == nullptr || ((C & 31) == 0 && (ldy3 & 31) == 0 && ldy3 >= C && (((size_t)y3) & 15) == 0), -3,
               "deft_upsample_add: y3 needs C %% 32 == 0 and ldy3 %% 32 == 0");
    DEFT_CHECK(x && wup && skip && y && (C & 3) == 0 && f >= 2 && (f & 1) == 0, -1, "deft_upsample_add: bad arguments (C=%d f=%d)", C, f);
    DEFT_CHECK((ldx & 3) == 0 && (lds & 3) == 0 && (ldy & 3) == 0, -2, "deft_upsample_add: ld %% 4 != 0");
    const long long tot = (long long)N * H * f * W * f * (C / 4);
    hipLaunchKernelGGL(upsample_add_kernel, dim3(deft_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, x, wup, skip, y, (deft_piece_t*)y3, N, H, W, C / 4, f, ldx, lds, ldy, ldy3);
    DEFT_CHECK_LAUNCH("upsample_add");
    return 0;
}

// ---------------------------------------------------------------------------
// sigmoid + 3x3 NMS + candidate compaction   (detector.py:488, utils.py:69-74)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// One 16x16 pixel tile of one (frame, class) per workgroup: the 18x18 neighbourhood is read and
// sigmoid'ed ONCE into LDS (the first version did 9 strided global reads + 9 expf per pixel).
__global__ __launch_bounds__(256) void hm_peaks_kernel(const float* __restrict__ hm, int N, int H, int W, int C, int ld, int sig,
                                                       float* __restrict__ cs, int* __restrict__ ci, int* __restrict__ cc, int cap) {
    __shared__ float t[18][19];
    const int tid = threadIdx.x;
    const int n = blockIdx.z / C, c = blockIdx.z - n * C;
    const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
    const float* base = hm + (size_t)n * H * W * ld + c;
    for (int i = tid; i < 18 * 18; i += 256) {
        const int ty = i / 18, tx = i - ty * 18;
        const int yy = y0 + ty - 1, xx = x0 + tx - 1;
        float v = -3.0e38f;                                      // outside the map: never blocks a peak
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            const float raw = base[(size_t)(yy * W + xx) * ld];
            v = sig ? sigmoidf_(raw) : raw;
        }
        t[ty][tx] = v;
    }
    __syncthreads();
    __shared__ int s_cnt, s_base;
    if (tid == 0) s_cnt = 0;
    const int ty = tid >> 4, tx = tid & 15;
    const int y = y0 + ty, x = x0 + tx;
    const float s = t[ty + 1][tx + 1];
    bool peak = y < H && x < W;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) peak = peak && (t[ty + dy][tx + dx] <= s);
    __syncthreads();
    const int slot = peak ? atomicAdd(&s_cnt, 1) : 0;           // rank inside the tile (LDS atomic)
    __syncthreads();
    if (tid == 0 && s_cnt > 0) s_base = atomicAdd(&cc[n], s_cnt);   // ONE global atomic per tile reserves the range
    __syncthreads();
    if (peak) {
        const int pos = s_base + slot;
        if (pos < cap) {
            cs[(size_t)n * cap + pos] = s;
            ci[(size_t)n * cap + pos] = c * H * W + y * W + x;
        }
    }
}

extern "C" int deft_hm_peaks(const float* hm, int N, int H, int W, int C, int ld, int apply_sigmoid,
                             float* cand_score, int* cand_idx, int* cand_count, int cap, void* stream) {
    DEFT_CHECK(hm && cand_score && cand_idx && cand_count && cap > 0 && ld >= C, -1, "deft_hm_peaks: bad arguments");
    DEFT_CHECK((long long)N * C <= 65535, -2, "deft_hm_peaks: N*C=%lld exceeds the grid limit", (long long)N * C);
    hipLaunchKernelGGL(hm_peaks_kernel, dim3(deft_cdiv(W, 16), deft_cdiv(H, 16), N * C), dim3(256), 0, (hipStream_t)stream,
                       hm, N, H, W, C, ld, apply_sigmoid, cand_score, cand_idx, cand_count, cap);
    DEFT_CHECK_LAUNCH("hm_peaks");
    return 0;
}

// ---------------------------------------------------------------------------
// top-K of the candidates: 64-bit keys (score bits : ~index) are all distinct, so a
// 6-pass MSB radix select finds the K-th key exactly; the K survivors are bitonic
// sorted in LDS (descending score, ascending index on ties).   (utils.py:89-104)
// ---------------------------------------------------------------------------
#define TOPK_MAXK 512

__device__ __forceinline__ unsigned long long topk_key(float s, int idx) {
    return ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
}

__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ cs, const int* __restrict__ ci,
                                                   const int* __restrict__ cc, int cap, int K, int HW,
                                                   float* __restrict__ out_s, int* __restrict__ out_i, int* __restrict__ out_c) {
    __shared__ int hist[2048];
    __shared__ int ssum[256];
    __shared__ unsigned long long sel[TOPK_MAXK];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_kth, s_nsel;
    const int n = blockIdx.x, tid = threadIdx.x;
    int cnt = cc[n];
    if (cnt > cap) cnt = cap;
    const float* s = cs + (size_t)n * cap;
    const int* ix = ci + (size_t)n * cap;
    for (int i = tid; i < TOPK_MAXK; i += 256) sel[i] = 0ull;
    if (tid == 0) { s_prefix = 0ull; s_kth = K; s_nsel = 0; }
    __syncthreads();
    if (cnt > K) {
        // passes of 11,11,11,11,11,9 bits from the top
        int hi = 64;
        for (int pass = 0; pass < 6; ++pass) {
            const int bits = pass < 5 ? 11 : 9;
            const int lo = hi - bits;
            for (int i = tid; i < 2048; i += 256) hist[i] = 0;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            for (int i = tid; i < cnt; i += 256) {
                const unsigned long long key = topk_key(s[i], ix[i]);
                if (pass == 0 || (key >> hi) == prefix) atomicAdd(&hist[(int)((key >> lo) & ((1u << bits) - 1))], 1);
            }
            __syncthreads();
            // locate the bin (from the top) where the running count reaches kth: every thread owns
            // 8 consecutive bins; inclusive suffix sums over the 256 owners (Hillis-Steele in LDS), then
            // the one owner whose range contains the crossing walks its 8 bins.
            {
                const int kth = s_kth;     // read before the barriers below: the owner rewrites it at the end
                int own = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) own += hist[tid * 8 + q];     // bins >= (1 << bits) are zero
                ssum[tid] = own;
                __syncthreads();
                for (int d = 1; d < 256; d <<= 1) {
                    const int add = tid + d < 256 ? ssum[tid + d] : 0;
                    __syncthreads();
                    ssum[tid] += add;
                    __syncthreads();
                }
                const int above = tid + 1 < 256 ? ssum[tid + 1] : 0;      // count in the bins above my range
                const bool mine = (ssum[tid] >= kth && above < kth) || (tid == 0 && ssum[0] < kth);
                if (mine) {
                    int rem = kth - above, b = tid * 8 + 7;
                    for (; b > tid * 8; --b) {
                        if (hist[b] >= rem) break;
                        rem -= hist[b];
                    }
                    s_kth = rem;
                    s_prefix = (prefix << bits) | (unsigned long long)b;
                }
            }
            __syncthreads();
            hi = lo;
        }
    }
    const unsigned long long thr = (cnt > K) ? s_prefix : 0ull;
    for (int i = tid; i < cnt; i += 256) {
        const unsigned long long key = topk_key(s[i], ix[i]);
        if (key >= thr) {
            const int pos = atomicAdd(&s_nsel, 1);
            if (pos < TOPK_MAXK) sel[pos] = key;
        }
    }
    __syncthreads();
    // bitonic sort, descending, TOPK_MAXK slots
    for (int k = 2; k <= TOPK_MAXK; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < TOPK_MAXK; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = sel[i], b = sel[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { sel[i] = b; sel[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int k = tid; k < K; k += 256) {
        const unsigned long long key = sel[k];
        float sc = 0.f; int ind = 0, cls = 0;
        if (key != 0ull) {
            sc = __uint_as_float((unsigned)(key >> 32));
            const unsigned full = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
            cls = (int)(full / (unsigned)HW);
            ind = (int)(full - (unsigned)cls * (unsigned)HW);
        }
        out_s[(size_t)n * K + k] = sc;
        out_i[(size_t)n * K + k] = ind;
        out_c[(size_t)n * K + k] = cls;
    }
}

extern "C" int deft_topk(const float* cand_score, const int* cand_idx, const int* cand_count,
                         int N, int cap, int K, int HW, float* out_score, int* out_ind, int* out_cls, void* stream) {
    DEFT_CHECK(cand_score && cand_idx && cand_count && out_score && out_ind && out_cls, -1, "deft_topk: null pointer");
    DEFT_CHECK(K > 0 && K <= TOPK_MAXK && HW > 0, -2, "deft_topk: K=%d must be in 1..%d", K, TOPK_MAXK);
    hipLaunchKernelGGL(topk_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, cand_score, cand_idx, cand_count, cap, K, HW, out_score, out_ind, out_cls);
    DEFT_CHECK_LAUNCH("topk");
    return 0;
}

// ---------------------------------------------------------------------------
// regression heads at the K peaks only   (base_model.py:37-66 + utils.py:32-36)
// ---------------------------------------------------------------------------
#define HP_PPB 4      // peaks per block
#define HP_MAXK 576   // 9 * 64

__global__ __launch_bounds__(256) void heads_at_peaks_kernel(const float* __restrict__ feat, int H, int W, int Cf, int ld,
                                                             const int* __restrict__ inds, int K,
                                                             const float* __restrict__ w0t, const float* __restrict__ b0,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             const int* __restrict__ head_of, int nheads, int Ctot,
                                                             float* __restrict__ out) {
    __shared__ float patch[HP_PPB][HP_MAXK];
    __shared__ float hid[HP_PPB][256];
    const int tid = threadIdx.x, n = blockIdx.y, k0 = blockIdx.x * HP_PPB;
    const int KK = 9 * Cf;
    for (int i = tid; i < HP_PPB * KK; i += 256) {
        const int pk = i / KK, kk = i - pk * KK;
        const int tap = kk / Cf, c = kk - tap * Cf;
        float v = 0.f;
        if (k0 + pk < K) {
            const int ind = inds[(size_t)n * K + k0 + pk];
            const int y = ind / W + tap / 3 - 1, x = ind % W + tap % 3 - 1;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = feat[((size_t)(n * H + y) * W + x) * ld + c];
        }
        patch[pk][kk] = v;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int h = 0; h < nheads; ++h) {
        float acc[HP_PPB];
        const float bb = b0[h * 256 + tid];
#pragma unroll
        for (int q = 0; q < HP_PPB; ++q) acc[q] = bb;
        const float* wp = w0t + (size_t)h * KK * 256 + tid;
        for (int kk = 0; kk < KK; ++kk) {
            const float w = wp[(size_t)kk * 256];
#pragma unroll
            for (int q = 0; q < HP_PPB; ++q) acc[q] += w * patch[q][kk];
        }
#pragma unroll
        for (int q = 0; q < HP_PPB; ++q) hid[q][tid] = fmaxf(acc[q], 0.f);
        __syncthreads();
        // wave w finishes peak w: out[c] = b2[c] + sum_o w2[c][o] * hid[w][o]
        for (int c = 0; c < Ctot; ++c) {
            if (head_of[c] != h) continue;
            float part = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) part += w2[(size_t)c * 256 + lane + 64 * q] * hid[wave][lane + 64 * q];
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (lane == 0 && k0 + wave < K) out[((size_t)n * K + k0 + wave) * Ctot + c] = part + b2[c];
        }
        __syncthreads();
    }
}

extern "C" int deft_heads_at_peaks(const float* feat, int N, int H, int W, int Cf, int ld,
                                   const int* inds, int K, const float* w0t, const float* b0,
                                   const float* w2, const float* b2, const int* head_of,
                                   int nheads, int Ctot, float* out, void* stream) {
    DEFT_CHECK(feat && inds && w0t && b0 && w2 && b2 && head_of && out, -1, "deft_heads_at_peaks: null pointer");
    DEFT_CHECK(9 * Cf <= HP_MAXK && nheads > 0 && Ctot > 0 && K > 0, -2, "deft_heads_at_peaks: Cf=%d too large or empty", Cf);
    hipLaunchKernelGGL(heads_at_peaks_kernel, dim3(deft_cdiv(K, HP_PPB), N), dim3(256), 0, (hipStream_t)stream,
                       feat, H, W, Cf, ld, inds, K, w0t, b0, w2, b2, head_of, nheads, Ctot, out);
    DEFT_CHECK_LAUNCH("heads_at_peaks");
    return 0;
}

// ---------------------------------------------------------------------------
// regression heads at the K peaks as a sparse-row conv GEMM: rows of the peaks (step 1), the
// 3x3 (Cf -> nheads*256) + bias + ReLU layer runs in deft_conv2d_nhwc with DeftGemmDesc.rowmap
// (step 2, MFMA), then the per-head 1x1 (256 -> c) here (step 3).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void peak_rows_kernel(const int* __restrict__ inds, int NK, int K, int H, int W, int* __restrict__ rowmap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NK) return;
    const int n = i / K, ind = inds[i];
    rowmap[2 * i] = n * H * W;
    rowmap[2 * i + 1] = ((ind / W) << 16) | (ind % W);
}

extern "C" int deft_peak_rows(const int* inds, int N, int K, int H, int W, int* rowmap, void* stream) {
    DEFT_CHECK(inds && rowmap && H > 0 && W > 0 && H < 65536 && W < 65536, -1, "deft_peak_rows: bad arguments");
    if (N * K <= 0) return 0;
    hipLaunchKernelGGL(peak_rows_kernel, dim3(deft_cdiv(N * K, 256)), dim3(256), 0, (hipStream_t)stream, inds, N * K, K, H, W, rowmap);
    DEFT_CHECK_LAUNCH("peak_rows");
    return 0;
}

// one wavefront per peak: out[i][c] = b2[c] + sum_o w2[c][o] * hid[i][head_of[c]*256 + o]
__global__ __launch_bounds__(256) void heads_finish_kernel(const float* __restrict__ hid, int ldh, int NK,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           const int* __restrict__ head_of, int Ctot, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= NK) return;
    const float* hp = hid + (size_t)i * ldh;
    for (int c = 0; c < Ctot; ++c) {
        const float* hv = hp + head_of[c] * 256;
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) part += w2[(size_t)c * 256 + lane + 64 * q] * hv[lane + 64 * q];
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) out[(size_t)i * Ctot + c] = part + b2[c];
    }
}

extern "C" int deft_heads_finish(const float* hid, int ldh, int NK, const float* w2, const float* b2, const int* head_of,
                                 int Ctot, float* out, void* stream) {
    DEFT_CHECK(hid && w2 && b2 && head_of && out && Ctot > 0 && ldh >= 256, -1, "deft_heads_finish: bad arguments");
    if (NK <= 0) return 0;
    hipLaunchKernelGGL(heads_finish_kernel, dim3(deft_cdiv(NK, 4)), dim3(256), 0, (hipStream_t)stream, hid, ldh, NK, w2, b2, head_of, Ctot, out);
    DEFT_CHECK_LAUNCH("heads_finish");
    return 0;
}

// ---------------------------------------------------------------------------
// box assembly   (decode.py:118-196)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_boxes_kernel(const int* __restrict__ inds, const float* __restrict__ heads,
                                                           int NK, int Wm, int Hm, int Ctot, int off_reg, int off_wh, int off_ltrb,
                                                           float* __restrict__ cts, float* __restrict__ bboxes,
                                                           float* __restrict__ centers) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NK) return;
    const int ind = inds[i];
    const float ys0 = (float)(ind / Wm), xs0 = (float)(ind % Wm);
    const float* hv = heads + (size_t)i * Ctot;
    cts[2 * i] = xs0; cts[2 * i + 1] = ys0;
    float xs = xs0 + 0.5f, ys = ys0 + 0.5f;
    if (off_reg >= 0) { xs = xs0 + hv[off_reg]; ys = ys0 + hv[off_reg + 1]; }
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (off_wh >= 0) {
        const float w = fmaxf(hv[off_wh], 0.f), h = fmaxf(hv[off_wh + 1], 0.f);
        b0 = xs - w / 2; b1 = ys - h / 2; b2 = xs + w / 2; b3 = ys + h / 2;
    }
    if (off_ltrb >= 0) {
        b0 = xs0 + hv[off_ltrb]; b1 = ys0 + hv[off_ltrb + 1]; b2 = xs0 + hv[off_ltrb + 2]; b3 = ys0 + hv[off_ltrb + 3];
    }
    bboxes[4 * i] = b0; bboxes[4 * i + 1] = b1; bboxes[4 * i + 2] = b2; bboxes[4 * i + 3] = b3;
    if (centers) {   // convert_detection: (2*x1/w + (x2-x1)/w) - 1
        centers[2 * i] = (2.f * (b0 / (float)Wm) + (b2 - b0) / (float)Wm) - 1.f;
        centers[2 * i + 1] = (2.f * (b1 / (float)Hm) + (b3 - b1) / (float)Hm) - 1.f;
    }
}

extern "C" int deft_decode_boxes(const int* inds, const float* heads, int N, int K, int Wm, int Hm, int Ctot,
                                 int off_reg, int off_wh, int off_ltrb_amodal, float* cts, float* bboxes, float* centers,
                                 void* stream) {
    DEFT_CHECK(inds && heads && cts && bboxes && Wm > 0, -1, "deft_decode_boxes: bad arguments");
    hipLaunchKernelGGL(decode_boxes_kernel, dim3(deft_cdiv(N * K, 256)), dim3(256), 0, (hipStream_t)stream,
                       inds, heads, N * K, Wm, Hm, Ctot, off_reg, off_wh, off_ltrb_amodal, cts, bboxes, centers);
    DEFT_CHECK_LAUNCH("decode_boxes");
    return 0;
}

// ---------------------------------------------------------------------------
// sparse embedding head for one feature map   (AFE.py:162-188)
// ---------------------------------------------------------------------------
#define EM_MAXC 512

// F.grid_sample's unnormalisation of a [-1,1] coordinate onto `size` pixels.  AFE.py:178 passes no align_corners:
// torch >= 1.3 (the oracle as run today) means False; the authors' torch 1.2 environment meant True.
__device__ __forceinline__ float grid_unnormalize(float g, int size, int align_corners) {
    return align_corners ? (g + 1.f) / 2.f * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) / 2.f;
}

// One workgroup per (detection, frame).  Threads = 4 bilinear corners x (Co/4) output quads x
// S k-slices: each thread accumulates 4 outputs over its slice of the 9*C contraction with
// float4 weight loads (a quad group reads Co*4 contiguous bytes per k), the slices are then
// reduced through LDS.  The 4x4xC input neighbourhood (zero padded = the conv's padding) is
// staged once in LDS.
__global__ __launch_bounds__(256) void embed_map_kernel(const float* __restrict__ fmap, int H, int W, int C, int ld,
                                                        const float* __restrict__ wsel_t, const float* __restrict__ bsel, int Co,
                                                        const float* __restrict__ centers, int ndet,
                                                        float* __restrict__ out, int ldo, int col_off, int align_corners) {
    __shared__ __attribute__((aligned(16))) float patch[16 * EM_MAXC];
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    __shared__ float vals[4][64];
    const int tid = threadIdx.x, i = blockIdx.x, n = blockIdx.y;
    const float gx = centers[((size_t)n * ndet + i) * 2], gy = centers[((size_t)n * ndet + i) * 2 + 1];
    // grid_sample: unnormalise, clip to the border, bilinear corners
    float fx = grid_unnormalize(gx, W, align_corners), fy = grid_unnormalize(gy, H, align_corners);
    fx = fminf((float)(W - 1), fmaxf(fx, 0.f));
    fy = fminf((float)(H - 1), fmaxf(fy, 0.f));
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int C4 = C >> 2;
    for (int t = tid; t < 16 * C4; t += 256) {
        const int pos = t / C4, c4 = t - pos * C4;
        const int py = y0 - 1 + (pos >> 2), px = x0 - 1 + (pos & 3);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W)
            v = *(const float4*)(fmap + ((size_t)(n * H + py) * W + px) * ld + 4 * c4);
        *(float4*)&patch[pos * C + 4 * c4] = v;
    }
    __syncthreads();
    const int G = Co >> 2;                 // output quads
    const int S = 256 / (4 * G);           // k-slices (Co=32 -> 8, 48 -> 5, 64 -> 4)
    const int og = tid % G, q = (tid / G) & 3, sl = tid / (4 * G);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sl < S) {
        const int qy = q >> 1, qx = q & 1;
        for (int tap = 0; tap < 9; ++tap) {
            const float* pp = &patch[((qy + tap / 3) * 4 + (qx + tap % 3)) * C];
            const float* wp = wsel_t + (size_t)tap * C * Co + 4 * og;
            for (int c = sl; c < C; c += S) {
                const float4 w = *(const float4*)(wp + (size_t)c * Co);
                const float xv = pp[c];
                acc.x += w.x * xv; acc.y += w.y * xv; acc.z += w.z * xv; acc.w += w.w * xv;
            }
        }
        *(float4*)&red[tid * 4] = acc;
    }
    __syncthreads();
    if (tid < 4 * Co) {                    // (corner q2, output o): sum the S slices, bias, ReLU
        const int q2 = tid / Co, o = tid - q2 * Co;
        float v = bsel[o];
        for (int s2 = 0; s2 < S; ++s2) v += red[((s2 * 4 + q2) * G + (o >> 2)) * 4 + (o & 3)];
        vals[q2][o] = fmaxf(v, 0.f);
    }
    __syncthreads();
    if (tid < Co) {
        const float x1f = x0f + 1.f, y1f = y0f + 1.f;
        const float nw = (x1f - fx) * (y1f - fy), ne = (fx - x0f) * (y1f - fy);
        const float sw = (x1f - fx) * (fy - y0f), se = (fx - x0f) * (fy - y0f);
        const bool xin = x0 + 1 <= W - 1, yin = y0 + 1 <= H - 1;
        float r = vals[0][tid] * nw;
        if (xin) r += vals[1][tid] * ne;
        if (yin) r += vals[2][tid] * sw;
        if (xin && yin) r += vals[3][tid] * se;
        out[((size_t)n * ndet + i) * ldo + col_off + tid] = r;
    }
}

extern "C" int deft_embed_map(const float* fmap, int Nf, int H, int W, int C, int ld,
                              const float* wsel_t, const float* bsel, int Co,
                              const float* centers, int ndet, float* out, int ldo, int col_off, int align_corners, void* stream) {
    DEFT_CHECK(fmap && wsel_t && bsel && centers && out, -1, "deft_embed_map: null pointer");
    DEFT_CHECK(C <= EM_MAXC && (C & 3) == 0 && (ld & 3) == 0 && Co <= 64 && Co >= 4 && (Co & 3) == 0, -2,
               "deft_embed_map: C=%d (<=%d, %%4) Co=%d (4..64, %%4)", C, EM_MAXC, Co);
    if (ndet <= 0) return 0;
    hipLaunchKernelGGL(embed_map_kernel, dim3(ndet, Nf), dim3(256), 0, (hipStream_t)stream,
                       fmap, H, W, C, ld, wsel_t, bsel, Co, centers, ndet, out, ldo, col_off, align_corners);
    DEFT_CHECK_LAUNCH("embed_map");
    return 0;
}

// ---------------------------------------------------------------------------
// fused embedding head, steps 1 and 3 (step 2 = grouped sparse-row conv GEMM, igemm.hip)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ centers, int Nf, int ndet,
                                                         const int* __restrict__ map_hw, int nmaps,
                                                         int* __restrict__ rowmap, float* __restrict__ bw, int align_corners) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int per = Nf * ndet;
    if (t >= nmaps * per) return;
    const int k = t / per, d = t - k * per;          // map, (frame, detection)
    const int n = d / ndet;
    const int H = map_hw[2 * k], W = map_hw[2 * k + 1];
    const float gx = centers[2 * d], gy = centers[2 * d + 1];
    // grid_sample: unnormalise, clip to the border, bilinear corners
    float fx = grid_unnormalize(gx, W, align_corners), fy = grid_unnormalize(gy, H, align_corners);
    fx = fminf((float)(W - 1), fmaxf(fx, 0.f));
    fy = fminf((float)(H - 1), fmaxf(fy, 0.f));
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float x1f = x0f + 1.f, y1f = y0f + 1.f;
    const bool xin = x0 + 1 <= W - 1, yin = y0 + 1 <= H - 1;
    const float w4[4] = {(x1f - fx) * (y1f - fy), xin ? (fx - x0f) * (y1f - fy) : 0.f,
                         yin ? (x1f - fx) * (fy - y0f) : 0.f, (xin && yin) ? (fx - x0f) * (fy - y0f) : 0.f};
    const bool ok4[4] = {true, xin, yin, xin && yin};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t r = ((size_t)k * per + d) * 4 + q;
        rowmap[2 * r] = n * H * W;
        rowmap[2 * r + 1] = ok4[q] ? (((y0 + (q >> 1)) << 16) | (x0 + (q & 1))) : -1;
        bw[r] = w4[q];
    }
}

extern "C" int deft_embed_rows(const float* centers, int Nf, int ndet, const int* map_hw, int nmaps,
                               int* rowmap, float* bw, int align_corners, void* stream) {
    DEFT_CHECK(centers && map_hw && rowmap && bw && nmaps > 0, -1, "deft_embed_rows: bad arguments");
    if (Nf * ndet <= 0) return 0;
    hipLaunchKernelGGL(embed_rows_kernel, dim3(deft_cdiv((long long)nmaps * Nf * ndet, 256)), dim3(256), 0, (hipStream_t)stream,
                       centers, Nf, ndet, map_hw, nmaps, rowmap, bw, align_corners);
    DEFT_CHECK_LAUNCH("embed_rows");
    return 0;
}

__global__ __launch_bounds__(256) void embed_blend_kernel(const float* __restrict__ tmp, const float* __restrict__ bw,
                                                          const int* __restrict__ map_out, int per,
                                                          float* __restrict__ out, int ldo) {
    const int k = blockIdx.y;
    const int toff = map_out[4 * k], ldt = map_out[4 * k + 1], Co = map_out[4 * k + 2], col = map_out[4 * k + 3];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= per * Co) return;
    const int d = t / Co, o = t - d * Co;
    const float* tp = tmp + (size_t)toff + (size_t)d * 4 * ldt + o;
    const float* w = bw + ((size_t)k * per + d) * 4;
    float r = tp[0] * w[0];
    r += tp[ldt] * w[1];
    r += tp[2 * ldt] * w[2];
    r += tp[3 * ldt] * w[3];
    out[(size_t)d * ldo + col + o] = r;
}

extern "C" int deft_embed_blend(const float* tmp, const float* bw, const int* map_out, int nmaps, int Nf, int ndet,
                                float* out, int ldo, void* stream) {
    DEFT_CHECK(tmp && bw && map_out && out && nmaps > 0, -1, "deft_embed_blend: bad arguments");
    if (Nf * ndet <= 0) return 0;
    hipLaunchKernelGGL(embed_blend_kernel, dim3(deft_cdiv((long long)Nf * ndet * 64, 256), nmaps), dim3(256), 0, (hipStream_t)stream,
                       tmp, bw, map_out, Nf * ndet, out, ldo);
    DEFT_CHECK_LAUNCH("embed_blend");
    return 0;
}

// ---------------------------------------------------------------------------
// affinity tail: final 64->1 + ReLU, dual softmax with analytic padding   (AFE.py:119-150)
// ---------------------------------------------------------------------------
#define AF_MAXOBJ 112   // 112*112*4 B = 49 KB of LDS (opts.py:339 max_object = 100)

// phase 1, every pair of every frame block in parallel: x = relu(h4 . w5 + b5), parked at the pair's final
// place in `out`.  16 lanes per pair row (one float4 each: a 256-byte row is one coalesced request), 4 rows per
// 16-lane group in flight, partial dot products combined with 4 xor-shuffles.
__global__ __launch_bounds__(256) void affinity_pairs_kernel(const float* __restrict__ h4, int ldh, int C4,
                                                             const float* __restrict__ w5, float b5, int TQ, int Q,
                                                             float* __restrict__ out) {
    const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    float acc[4];
    int pid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = (blockIdx.x * 4 + u) * 16 + grp;
        pid[u] = p;
        acc[u] = 0.f;
        if (p < TQ) {
            const float* hp = h4 + (size_t)p * ldh;
            for (int k = sub * 4; k < C4; k += 64) {
                const float4 v = *(const float4*)(hp + k);
                acc[u] += v.x * w5[k] + v.y * w5[k + 1] + v.z * w5[k + 2] + v.w * w5[k + 3];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float a = acc[u];
        a += __shfl_xor(a, 8);
        a += __shfl_xor(a, 4);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 1);
        if (sub == 0 && pid[u] < TQ) {
            const int t = pid[u] / Q, j = pid[u] - t * Q;
            out[(size_t)t * (Q + 1) + j] = fmaxf(a + b5, 0.f);          // the relu'd logit; the softmaxes of phase 2 subtract their maxima
        }
    }
}

// phase 2, one block per history frame: row / column sums with the analytic padding terms, max of the two
// softmaxes, unmatched column -- in place on `out`.
__global__ __launch_bounds__(256) void affinity_finish_kernel(const int* __restrict__ row_start, int Q, int max_object,
                                                              float* __restrict__ out) {
    __shared__ float E[AF_MAXOBJ * AF_MAXOBJ];
    __shared__ float rs[AF_MAXOBJ], csum[AF_MAXOBJ], rmx[AF_MAXOBJ], cmx[AF_MAXOBJ];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int t0 = row_start[f], P = row_start[f + 1] - t0;
    if (P < 0 || P > AF_MAXOBJ || P > max_object) {
        // row_start is DEVICE data the host entry cannot validate: a frame with more objects than the tile holds is refused
        // loudly (NaN rows) instead of overrunning LDS
        for (int p = tid; p < (P > 0 ? P : 0) * (Q + 1); p += 256) out[(size_t)t0 * (Q + 1) + p] = __int_as_float(0x7fc00000);
        return;
    }
    for (int p = tid; p < P * Q; p += 256) {
        const int i = p / Q, j = p - i * Q;
        E[p] = out[(size_t)(t0 + i) * (Q + 1) + j];
    }
    __syncthreads();
    // F.softmax (AFE.py:136-137) subtracts the maximum of the softmax dimension: every row / column also holds the constant 1.0 of the
    // appended unmatched entry and (max_object - n) zero paddings, so its maximum is at least 1 -- exp never overflows, whatever the logits
    if (tid < P) {
        float m = 1.f;
        for (int j = 0; j < Q; ++j) m = fmaxf(m, E[tid * Q + j]);
        float s = 0.f;
        for (int j = 0; j < Q; ++j) s += expf(E[tid * Q + j] - m);
        rmx[tid] = m;
        rs[tid] = s + (float)(max_object - Q) * expf(-m) + expf(1.f - m);
    }
    if (tid < Q) {
        float m = 1.f;
        for (int i = 0; i < P; ++i) m = fmaxf(m, E[i * Q + tid]);
        float s = 0.f;
        for (int i = 0; i < P; ++i) s += expf(E[i * Q + tid] - m);
        cmx[tid] = m;
        csum[tid] = s + (float)(max_object - P) * expf(-m) + expf(1.f - m);
    }
    __syncthreads();
    for (int p = tid; p < P * (Q + 1); p += 256) {
        const int i = p / (Q + 1), j = p - i * (Q + 1);
        float v;
        if (j < Q) {
            const float x = E[i * Q + j];
            v = fmaxf(expf(x - rmx[i]) / rs[i], expf(x - cmx[j]) / csum[j]);
        } else {
            v = expf(1.f - rmx[i]) / rs[i];
        }
        out[(size_t)(t0 + i) * (Q + 1) + j] = v;
    }
}

extern "C" int deft_affinity_finish(const float* h4, int ldh, int C4, const float* w5, float b5,
                                    const int* row_start, int F, int T, int Q, int max_object, float* out, void* stream) {
    DEFT_CHECK((h4 == nullptr || w5) && row_start && out, -1, "deft_affinity_finish: null pointer");
    DEFT_CHECK(Q > 0 && Q <= AF_MAXOBJ && max_object <= AF_MAXOBJ && Q <= max_object && (C4 & 3) == 0 && (ldh & 3) == 0, -2,
               "deft_affinity_finish: Q=%d max_object=%d must be <= %d", Q, max_object, AF_MAXOBJ);
    if (F <= 0) return 0;
    DEFT_CHECK(T > 0 && (long long)T * Q < (1ll << 31), -3, "deft_affinity_finish: T=%d history rows in total (= row_start[F])", T);
    if (h4 != nullptr) {                                             // (h4 == NULL: deft_pair_mlp has left the relu'd logits at their places in `out`)
        hipLaunchKernelGGL(affinity_pairs_kernel, dim3(deft_cdiv((long long)T * Q, 64)), dim3(256), 0, (hipStream_t)stream, h4, ldh, C4, w5, b5, T * Q, Q, out);
        DEFT_CHECK_LAUNCH("affinity_pairs");
    }
    hipLaunchKernelGGL(affinity_finish_kernel, dim3(F), dim3(256), 0, (hipStream_t)stream, row_start, Q, max_object, out);
    DEFT_CHECK_LAUNCH("affinity_finish");
    return 0;
}

// ---------------------------------------------------------------------------
// batched single-step LSTM + 2 Linears   (kalman_filter_lstm.py:9-29, 65-78)
// ---------------------------------------------------------------------------
// One block of 512 threads = one track.  xin [nin] / hin [128] are already in LDS; on return o2 (LDS, nout
// values) holds the second Linear's output; h, c rows are updated in place.
struct LstmW {
    const float *wih_t, *whh_t, *bias, *w1_t, *b1, *w2_t, *b2;
};
__device__ __forceinline__ void lstm_block(const float* xin, const float* hin, float* gates, float* hnew, float* o1, float* o2,
                                           float* __restrict__ hrow, float* __restrict__ crow, int nin, int nout, const LstmW& w) {
    const int tid = threadIdx.x;
    float acc = w.bias[tid];
    for (int k = 0; k < nin; ++k) acc += w.wih_t[k * 512 + tid] * xin[k];
    for (int k = 0; k < 128; ++k) acc += w.whh_t[k * 512 + tid] * hin[k];
    gates[tid] = acc;
    __syncthreads();
    if (tid < 128) {
        const float ig = sigmoidf_(gates[tid]), fg = sigmoidf_(gates[128 + tid]);
        const float gg = tanhf(gates[256 + tid]), og = sigmoidf_(gates[384 + tid]);
        const float cn = fg * crow[tid] + ig * gg;
        const float hn = og * tanhf(cn);
        crow[tid] = cn;
        hrow[tid] = hn;
        hnew[tid] = hn;
    }
    __syncthreads();
    if (tid < 64) {
        float a = w.b1[tid];
        for (int k = 0; k < 128; ++k) a += w.w1_t[k * 64 + tid] * hnew[k];
        o1[tid] = a;
    }
    __syncthreads();
    if (tid < nout) {
        float a = w.b2[tid];
        for (int k = 0; k < 64; ++k) a += w.w2_t[k * nout + tid] * o1[k];
        o2[tid] = a;
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void lstm_step_kernel(const float* __restrict__ x, float* __restrict__ h, float* __restrict__ c,
                                                        int nin, int nout, LstmW w, float* __restrict__ pred) {
    __shared__ float xin[32], hin[128], gates[512], hnew[128], o1[64], o2[64];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (tid < nin) xin[tid] = x[(size_t)t * nin + tid];
    if (tid < 128) hin[tid] = h[(size_t)t * 128 + tid];
    __syncthreads();
    lstm_block(xin, hin, gates, hnew, o1, o2, h + (size_t)t * 128, c + (size_t)t * 128, nin, nout, w);
    if (tid < nout) pred[(size_t)t * nout + tid] = o2[tid];
}

extern "C" int deft_lstm_step(const float* x, float* h, float* c, int T, int nin, int nout,
                              const float* wih_t, const float* whh_t, const float* bias,
                              const float* w1_t, const float* b1, const float* w2_t, const float* b2,
                              float* pred, void* stream) {
    DEFT_CHECK(x && h && c && wih_t && whh_t && bias && w1_t && b1 && w2_t && b2 && pred, -1, "deft_lstm_step: null pointer");
    DEFT_CHECK(nin > 0 && nin <= 32 && nout > 0 && nout <= 64, -2, "deft_lstm_step: nin=%d (<=32) nout=%d (<=64)", nin, nout);
    if (T <= 0) return 0;
    const LstmW w = {wih_t, whh_t, bias, w1_t, b1, w2_t, b2};
    hipLaunchKernelGGL(lstm_step_kernel, dim3(T), dim3(512), 0, (hipStream_t)stream, x, h, c, nin, nout, w, pred);
    DEFT_CHECK_LAUNCH("lstm_step");
    return 0;
}

// ---------------------------------------------------------------------------
// fused motion-model update: feature builder + LSTM step + future boxes
// (tracker.py:408-480 / 482-580, kalman_filter_lstm.py:65-78)
// ---------------------------------------------------------------------------
// last [S][DEFT_MOTION_LAST] doubles per track slot: [0] 0 = no previous observation (STrack.first_time), 1 = has one;
// [1] last_frame_id; 2-D: [2] last_cx [3] last_cy [4] last_w(=tlwh[2]) [5] last_h(=tlwh[3]);
// 3-D: [2] last_h [3] last_w [4] last_l [5] last_cx [6] last_cy [7] last_cz [8] last_rot_y.
__global__ __launch_bounds__(512) void motion_step_kernel(const int* __restrict__ slot, const double* __restrict__ box, int dim, int frame_id,
                                                          float* __restrict__ h, float* __restrict__ c, double* __restrict__ last,
                                                          int nin, int nout, LstmW w, float* __restrict__ feat, double* __restrict__ pred) {
    __shared__ float xin[32], hin[128], gates[512], hnew[128], o1[64], o2[64];
    __shared__ double bx[8];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int s = slot[t];
    double* L = last + (size_t)s * DEFT_MOTION_LAST;
    if (tid < dim) bx[tid] = box[(size_t)t * dim + tid];
    if (tid < 128) hin[tid] = h[(size_t)s * 128 + tid];
    __syncthreads();
    if (tid == 0) {
        // every operation below is the reference's float64 Python arithmetic in the same order; the
        // features are rounded to float32 once at the end (`.float()`, tracker.py:465 / 569)
        const bool first = L[0] == 0.0;
        const double dt = (double)frame_id - L[1];
        double f[18];
        if (dim == 4) {
            const double cx = bx[0] + bx[2] / 2, cy = bx[1] + bx[3] / 2, bw = bx[2], bh = bx[3];
            f[0] = cx; f[1] = cy;
            f[2] = first ? 0.0 : (cx - L[2]) / dt; f[3] = first ? 0.0 : (cy - L[3]) / dt;
            f[4] = bh; f[5] = bw; f[6] = bw / bh;
            f[7] = first ? 0.0 : bh - L[5]; f[8] = first ? 0.0 : bw - L[4];
            f[9] = f[2]; f[10] = f[3];
            L[2] = cx; L[3] = cy; L[4] = bw; L[5] = bh;
        } else {
            const double bh = bx[0], bw = bx[1], bl = bx[2], cx = bx[3], cy = bx[4], cz = bx[5], rot = bx[6];
            f[0] = cx; f[1] = cy; f[2] = cz;
            f[3] = first ? 0.0 : cx - L[5]; f[4] = first ? 0.0 : cy - L[6]; f[5] = first ? 0.0 : cz - L[7];
            f[6] = bh; f[7] = bw; f[8] = bl;
            f[9] = first ? 0.0 : bh - L[2]; f[10] = first ? 0.0 : bw - L[3]; f[11] = first ? 0.0 : bl - L[4];
            f[12] = first ? 0.0 : (cx - L[5]) / dt; f[13] = first ? 0.0 : (cy - L[6]) / dt; f[14] = first ? 0.0 : (cz - L[7]) / dt;
            f[15] = rot; f[16] = first ? 0.0 : rot - L[8]; f[17] = first ? 0.0 : (rot - L[8]) / dt;
            L[2] = bh; L[3] = bw; L[4] = bl; L[5] = cx; L[6] = cy; L[7] = cz; L[8] = rot;
        }
        L[0] = 1.0; L[1] = (double)frame_id;
        for (int k = 0; k < nin; ++k) {
            xin[k] = (float)f[k];
            feat[(size_t)t * nin + k] = xin[k];
        }
    }
    __syncthreads();
    lstm_block(xin, hin, gates, hnew, o1, o2, h + (size_t)s * 128, c + (size_t)s * 128, nin, nout, w);
    const int nfut = nout / 4;
    if (tid < nfut) {
        // numpy in-place `float32 += float64`: the sum is formed in float64 and stored as float32
        const float* a = o2 + tid * 4;
        if (dim == 4) {
            const double cx = bx[0] + bx[2] / 2, cy = bx[1] + bx[3] / 2;
            const float p0 = (float)((double)a[0] + cx), p1 = (float)((double)a[1] + cy);
            const float ph = (float)((double)a[2] + bx[3]), pw = (float)((double)a[3] + bx[2]);
            double* o = pred + ((size_t)t * nfut + tid) * 4;           // xyah: (cx, cy, w/h, h)
            o[0] = p0; o[1] = p1; o[2] = pw / ph; o[3] = ph;
        } else {
            double* o = pred + ((size_t)t * nfut + tid) * 7;           // (h, w, l, x, y, z, rot_y)
            o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2];
            o[3] = (float)((double)a[0] + bx[3]); o[4] = (float)((double)a[1] + bx[4]);
            o[5] = (float)((double)a[2] + bx[5]); o[6] = (float)((double)a[3] + bx[6]);
        }
    }
}

extern "C" int deft_motion_step(const int* slot, const double* box, int T, int dim, int frame_id,
                                float* h, float* c, double* last, int nin, int nout,
                                const float* wih_t, const float* whh_t, const float* bias,
                                const float* w1_t, const float* b1, const float* w2_t, const float* b2,
                                float* feat, double* pred, void* stream) {
    DEFT_CHECK(slot && box && h && c && last && wih_t && whh_t && bias && w1_t && b1 && w2_t && b2 && feat && pred, -1,
               "deft_motion_step: null pointer");
    DEFT_CHECK((dim == 4 && nin == 11) || (dim == 7 && nin == 18), -2,
               "deft_motion_step: dim=%d nin=%d (tlwh: 4/11, 3-D box: 7/18)", dim, nin);
    DEFT_CHECK(nout > 0 && nout <= 64 && nout % 4 == 0, -2, "deft_motion_step: nout=%d (multiple of 4, <= 64)", nout);
    if (T <= 0) return 0;
    const LstmW w = {wih_t, whh_t, bias, w1_t, b1, w2_t, b2};
    hipLaunchKernelGGL(motion_step_kernel, dim3(T), dim3(512), 0, (hipStream_t)stream, slot, box, dim, frame_id, h, c, last, nin, nout, w, feat, pred);
    DEFT_CHECK_LAUNCH("motion_step");
    return 0;
}

// ---------------------------------------------------------------------------
// track x detection similarity of one frame (tracker.py:219-252, 663-688)
// ---------------------------------------------------------------------------
#define TS_MAXNODES 8
__global__ __launch_bounds__(128) void track_similarity_kernel(const float* __restrict__ sim, int ld, const int* __restrict__ node_row,
                                                               const float* __restrict__ node_scale, const int* __restrict__ node_cnt,
                                                               int L, float* __restrict__ out) {
    const int t = blockIdx.x;
    const int n = node_cnt[t];
    for (int j = threadIdx.x; j < ld; j += 128) {
        float v[TS_MAXNODES];
#pragma unroll
        for (int i = 0; i < TS_MAXNODES; ++i)
            v[i] = i < n ? sim[(size_t)node_row[t * L + i] * ld + j] * node_scale[t * L + i] : INFINITY;
#pragma unroll
        for (int a = 0; a < TS_MAXNODES - 1; ++a)
#pragma unroll
            for (int b = 0; b < TS_MAXNODES - 1 - a; ++b) {
                const float lo = fminf(v[b], v[b + 1]), hi = fmaxf(v[b], v[b + 1]);
                v[b] = lo; v[b + 1] = hi;
            }
        // numpy.median: the middle value, or the float32 mean of the two middle values
        const int k1 = n >> 1, k0 = (n & 1) ? k1 : k1 - 1;
        float m0 = 0.f, m1 = 0.f;
#pragma unroll
        for (int i = 0; i < TS_MAXNODES; ++i) {
            if (i == k0) m0 = v[i];
            if (i == k1) m1 = v[i];
        }
        out[(size_t)t * ld + j] = n == 0 ? 0.f : (n & 1) ? m1 : (m0 + m1) / 2.f;
    }
}

extern "C" int deft_track_similarity(const float* sim, int rows, int Q, const int* node_row, const float* node_scale,
                                     const int* node_cnt, int T, int L, float* out, void* stream) {
    DEFT_CHECK(sim && node_row && node_scale && node_cnt && out, -1, "deft_track_similarity: null pointer");
    DEFT_CHECK(rows > 0 && Q > 0 && L > 0 && L <= TS_MAXNODES, -2, "deft_track_similarity: rows=%d Q=%d L=%d (L <= %d)", rows, Q, L, TS_MAXNODES);
    if (T <= 0) return 0;
    hipLaunchKernelGGL(track_similarity_kernel, dim3(T), dim3(128), 0, (hipStream_t)stream, sim, Q + 1, node_row, node_scale, node_cnt, L, out);
    DEFT_CHECK_LAUNCH("track_similarity");
    return 0;
}

// ---- sum of the partial maps of a folded 1x1 conv (DeftGemmDesc.fold_y) + bias ----
__global__ __launch_bounds__(256) void fold_finish_kernel(const float* __restrict__ part, int nparts, long long M, int C, int ldp, const float* __restrict__ bias,
                                                          float* __restrict__ y, int ldy) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * C) return;
    const long long m = i / C;
    const int c = (int)(i - m * C);
    float s = part[m * ldp + c];
    for (int k = 1; k < nparts; ++k) s += part[((long long)k * M + m) * ldp + c];
    y[m * ldy + c] = s + (bias ? bias[c] : 0.f);
}

extern "C" int deft_fold_finish(const float* part, int nparts, long long M, int C, int ldp, const float* bias, float* y, int ldy, void* stream) {
    DEFT_CHECK(part && y && nparts >= 1 && M > 0 && C >= 1 && ldp >= C && ldy >= C, -1, "deft_fold_finish: bad arguments");
    DEFT_CHECK(M * C < (1ll << 31) * 256, -2, "deft_fold_finish: too many elements");
    hipLaunchKernelGGL(fold_finish_kernel, dim3((unsigned)deft_cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, part, nparts, M, C, ldp, bias, y, ldy);
    DEFT_CHECK_LAUNCH("fold_finish");
    return 0;
}
