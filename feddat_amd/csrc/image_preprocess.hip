// Input pipeline, image half: what the reference does on the host, three times per batch,
// in `ViltEncoderWrapper.process_inputs` (src/modeling/vilt.py:87-100) through HF `ViltProcessor` ->
// `ViltImageProcessor` (transformers; not vendored): PIL BICUBIC resize of the uint8 image to the ViLT size rule
// (shorter edge 384, longer <= 640, both floored to multiples of 32 -- computed by the caller), rescale 1/255,
// normalise with mean = std = 0.5, zero-pad to the batch maximum, pixel_mask.
//
// Bit-exact with Pillow's 8-bit `ImagingResample` (src/libImaging/Resample.c): separable convolution, horizontal pass
// first, bicubic kernel (a = -0.5) with its support widened by the down-scaling factor, coefficients normalised in
// double precision (no FMA contraction: the table kernel is compiled with contraction off) and quantised to 22
// fractional bits, int32 accumulation from 1 << 21, arithmetic shift, clip to [0, 255] after each pass.
// HBM-bound byte work: one thread per output pixel, 3 channels, taps read straight from the (L2-resident) source rows.
#include "common.hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int MAX_IMAGES = 48;           // per launch (descriptor table travels as a kernel argument)

struct ImgDesc {
    long src_off;        // byte offset of the packed [h][w][3] uint8 image in `images`
    long tmp_off;        // byte offset of the [h][ow][3] uint8 intermediate in the workspace
    long tab_off[2];     // byte offsets of the coefficient tables (axis 0 = horizontal, 1 = vertical)
    int h, w, oh, ow;
    int ksize[2];
};
struct ImgBatch {
    ImgDesc d[MAX_IMAGES];
    int n;
};

// table layout per axis: [out][2] int32 bounds (xmin, count) followed by [out][ksize] int32 coefficients
#pragma clang fp contract(off)
__device__ double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void coeff_kernel(ImgBatch B, char* __restrict__ ws) {
    const int img = blockIdx.y, axis = blockIdx.z;
    if (img >= B.n) return;
    const ImgDesc& d = B.d[img];
    const int in_size = axis == 0 ? d.w : d.h, out_size = axis == 0 ? d.ow : d.oh;
    const int xx = blockIdx.x * 64 + threadIdx.x;
    if (xx >= out_size) return;
    const int ksize = d.ksize[axis];
    int* bounds = reinterpret_cast<int*>(ws + d.tab_off[axis]);
    int* kk = bounds + 2 * out_size + (size_t)xx * ksize;
    double scale = (double)in_size / (double)out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = ((double)xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_filter(((double)(x + xmin) - center + 0.5) * ss);
    for (int x = 0; x < xmax; ++x) {
        double v = bicubic_filter(((double)(x + xmin) - center + 0.5) * ss);
        if (ww != 0.0) v /= ww;
        kk[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << PRECISION_BITS)) : (int)(0.5 + v * (double)(1 << PRECISION_BITS));
    }
    for (int x = xmax; x < ksize; ++x) kk[x] = 0;
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass: src [h][w][3] -> tmp [h][ow][3]
__global__ __launch_bounds__(256) void hpass_kernel(ImgBatch B, const uint8_t* __restrict__ images,
                                                    char* __restrict__ ws) {
    const int img = blockIdx.y;
    if (img >= B.n) return;
    const ImgDesc& d = B.d[img];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)d.h * d.ow) return;
    const int y = (int)(i / d.ow), xx = (int)(i - (long)y * d.ow);
    const int* bounds = reinterpret_cast<const int*>(ws + d.tab_off[0]);
    const int* kk = bounds + 2 * d.ow + (size_t)xx * d.ksize[0];
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const uint8_t* row = images + d.src_off + ((size_t)y * d.w + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int k = kk[x];
        s0 += row[3 * x] * k;
        s1 += row[3 * x + 1] * k;
        s2 += row[3 * x + 2] * k;
    }
    uint8_t* o = reinterpret_cast<uint8_t*>(ws + d.tmp_off) + ((size_t)y * d.ow + xx) * 3;
    o[0] = (uint8_t)clip8(s0 >> PRECISION_BITS);
    o[1] = (uint8_t)clip8(s1 >> PRECISION_BITS);
    o[2] = (uint8_t)clip8(s2 >> PRECISION_BITS);
}

// vertical pass + rescale + normalise + pad + mask: tmp [h][ow][3] -> pixel_values [B][3][Hm][Wm], pixel_mask [B][Hm][Wm]
__global__ __launch_bounds__(256) void vpass_kernel(ImgBatch B, const char* __restrict__ ws, int img0, int Hm, int Wm,
                                                    float* __restrict__ px, long* __restrict__ pmask) {
    const int img = blockIdx.y;
    if (img >= B.n) return;
    const ImgDesc& d = B.d[img];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Hm * Wm) return;
    const int yy = (int)(i / Wm), x = (int)(i - (long)yy * Wm);
    const size_t plane = (size_t)Hm * Wm;
    float* o = px + (size_t)(img0 + img) * 3 * plane + i;
    long* m = pmask ? pmask + (size_t)(img0 + img) * plane + i : nullptr;
    if (yy >= d.oh || x >= d.ow) {
        o[0] = 0.f; o[plane] = 0.f; o[2 * plane] = 0.f;
        if (m) *m = 0;
        return;
    }
    const int* bounds = reinterpret_cast<const int*>(ws + d.tab_off[1]);
    const int* kk = bounds + 2 * d.oh + (size_t)yy * d.ksize[1];
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const uint8_t* col = reinterpret_cast<const uint8_t*>(ws + d.tmp_off) + ((size_t)ymin * d.ow + x) * 3;
    const size_t rs = (size_t)d.ow * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
        const int k = kk[t];
        s0 += col[t * rs] * k;
        s1 += col[t * rs + 1] * k;
        s2 += col[t * rs + 2] * k;
    }
    const int r[3] = {clip8(s0 >> PRECISION_BITS), clip8(s1 >> PRECISION_BITS), clip8(s2 >> PRECISION_BITS)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // transformers rescale: float64 product rounded to float32; normalize: float32 (v - 0.5) / 0.5
        const float v = (float)((double)r[c] * (1.0 / 255.0));
        o[c * plane] = (v - 0.5f) / 0.5f;
    }
    if (m) *m = 1;
}

int ksize_for(int in_size, int out_size) {
    double fs = (double)in_size / (double)out_size;
    if (fs < 1.0) fs = 1.0;
    const double support = 2.0 * fs;
    return (int)ceil(support) * 2 + 1;
}

long align16(long v) { return (v + 15) & ~15L; }

// fills the descriptors; returns the workspace bytes needed (or -1)
long plan(const long* offsets, const int* h, const int* w, const int* oh, const int* ow, int n, ImgDesc* d) {
    long off = 0;
    for (int i = 0; i < n; ++i) {
        if (h[i] <= 0 || w[i] <= 0 || oh[i] <= 0 || ow[i] <= 0) return -1;
        ImgDesc e;
        e.src_off = offsets ? offsets[i] : 0;
        e.h = h[i]; e.w = w[i]; e.oh = oh[i]; e.ow = ow[i];
        e.ksize[0] = ksize_for(w[i], ow[i]);
        e.ksize[1] = ksize_for(h[i], oh[i]);
        e.tmp_off = off;
        off = align16(off + (long)h[i] * ow[i] * 3);
        e.tab_off[0] = off;
        off = align16(off + (long)ow[i] * (2 + e.ksize[0]) * 4);
        e.tab_off[1] = off;
        off = align16(off + (long)oh[i] * (2 + e.ksize[1]) * 4);
        if (d) d[i] = e;
    }
    return off;
}

}  // namespace

extern "C" long feddat_vilt_image_workspace_bytes(const int* heights, const int* widths, const int* out_h,
                                                  const int* out_w, int n) {
    if (!heights || !widths || !out_h || !out_w || n <= 0) return -1;
    return plan(nullptr, heights, widths, out_h, out_w, n, nullptr);
}

extern "C" int feddat_vilt_image_preprocess(const uint8_t* images, const long* offsets, const int* heights,
                                            const int* widths, const int* out_h, const int* out_w, int n, int Hm,
                                            int Wm, float* pixel_values, long* pixel_mask, void* workspace,
                                            long workspace_bytes, hipStream_t stream) {
    FD_CHECK_ARG(images && offsets && heights && widths && out_h && out_w && n > 0 && pixel_values && workspace);
    FD_CHECK_ARG(Hm > 0 && Wm > 0);
    for (int i = 0; i < n; ++i) FD_CHECK_ARG(out_h[i] <= Hm && out_w[i] <= Wm);
    FD_CHECK_ARG(plan(nullptr, heights, widths, out_h, out_w, n, nullptr) <= workspace_bytes);
    // the plan of the whole batch fixes the workspace offsets; launches go out in groups of MAX_IMAGES descriptors
    long base = 0;
    for (int i0 = 0; i0 < n; i0 += MAX_IMAGES) {
        const int m = n - i0 < MAX_IMAGES ? n - i0 : MAX_IMAGES;
        ImgBatch B;
        B.n = m;
        const long used = plan(offsets + i0, heights + i0, widths + i0, out_h + i0, out_w + i0, m, B.d);
        if (used < 0) return FEDDAT_EINVAL;
        int max_out = 0;
        long max_h = 0;
        for (int i = 0; i < m; ++i) {
            B.d[i].tmp_off += base; B.d[i].tab_off[0] += base; B.d[i].tab_off[1] += base;
            max_out = max_out > out_h[i0 + i] ? max_out : out_h[i0 + i];
            max_out = max_out > out_w[i0 + i] ? max_out : out_w[i0 + i];
            const long e = (long)heights[i0 + i] * out_w[i0 + i];
            max_h = max_h > e ? max_h : e;
        }
        base += used;
        char* ws = static_cast<char*>(workspace);
        hipLaunchKernelGGL(coeff_kernel, dim3((max_out + 63) / 64, m, 2), dim3(64), 0, stream, B, ws);
        hipLaunchKernelGGL(hpass_kernel, dim3((unsigned)((max_h + 255) / 256), m), dim3(256), 0, stream, B, images, ws);
        hipLaunchKernelGGL(vpass_kernel, dim3((unsigned)(((long)Hm * Wm + 255) / 256), m), dim3(256), 0, stream, B,
                           (const char*)ws, i0, Hm, Wm, pixel_values, pixel_mask);
    }
    FD_LAUNCH_RET();
}
