// Geometric augmentation on device: per-sample crop followed by horizontal / vertical flip, one pass,
// replacing Transforms.crop + horizontal_flip + vertical_flip (src/transforms.py:337-407, 955-1034).  The random draws stay
// with the caller (the reference draws them with torch's generator, :338-350, :391-403); this kernel is the data movement:
//   dst[b, c, y, x] = src[b, c, start_y[b] + (vflip[b] ? h-1-y : y), start_x[b] + (hflip[b] ? w-1-x : x)]
// HBM-bound copy: 4 B read + 4 B written per output element; rows are read forwards or backwards, both coalesced.
#include "ptta_common.h"
#include "ptta_kernels.h"
#include "../../include/ptta.h"

__global__ void crop_flip_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int H, int W, int h, int w,
                                 const int* __restrict__ start_y, const int* __restrict__ start_x,
                                 const unsigned char* __restrict__ hflip, const unsigned char* __restrict__ vflip) {
    const int b = blockIdx.z, row = blockIdx.y;                // row = channel * h + y
    const int ch = row / h, y = row - ch * h;
    int y0 = start_y ? start_y[b] : 0, x0 = start_x ? start_x[b] : 0;
    y0 = min(max(y0, 0), H - h); x0 = min(max(x0, 0), W - w);   // an out-of-range start never reads outside the sample
    const bool hf = hflip && hflip[b], vf = vflip && vflip[b];
    const float* s = src + (((size_t)b * c + ch) * H + y0 + (vf ? h - 1 - y : y)) * W + x0;
    float* d = dst + (((size_t)b * c + ch) * h + y) * w;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x)
        d[x] = s[hf ? w - 1 - x : x];
}

extern "C" int ptta_crop_flip(const float* src, float* dst, int n, int channels, int height, int width, int crop_height,
                              int crop_width, const int32_t* start_y, const int32_t* start_x, const uint8_t* hflip,
                              const uint8_t* vflip, ptta_stream stream) {
    if (!src || !dst || n <= 0 || channels <= 0 || crop_height <= 0 || crop_width <= 0 || crop_height > height ||
        crop_width > width || (long)channels * crop_height > 65535 || n > 65535)
        return -22;
    if (src == dst) return -22;                                // not an in-place operation
    dim3 grid((crop_width + 255) / 256, channels * crop_height, n);
    hipLaunchKernelGGL(crop_flip_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, channels, height, width,
                       crop_height, crop_width, start_y, start_x, hflip, vflip);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---- rotation, resize-and-crop, photometric jitter: the augmentations every adapt script enables (bash/adapt/adapt_msgchn_vkitti.sh:
// 34-41: rotate <= 5 degrees, resize_and_crop 1.0-1.5, brightness / contrast / saturation 0.6-1.4), applied inside every step
// (src/tta_main.py:595-605).  The reference implements them with torchvision.transforms.functional (absent from this image and from
// the reference tree: PARITY UNPINNED); the kernels restate torchvision 0.10.1's published tensor algorithms, and the oracle
// (oracle/transforms_oracle.py) restates them with the torch primitives torchvision itself calls (grid_sample, interpolate).

// Transforms.rotate (src/transforms.py:1036-1070) -> functional.rotate(image, angle, interpolation, expand=False) on a tensor:
// inverse matrix [cos a, -sin a, 0; sin a, cos a, 0] about the image centre, affine grid over pixel centres
// (x_c = ox + 0.5 - W/2), grid_sample(align_corners=False, padding zeros), nearest (round half to even) or bilinear.
__global__ void rotate_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int H, int W,
                              const unsigned char* __restrict__ do_rotate, const float* __restrict__ angle_deg, int bilinear) {
    const int b = blockIdx.z;
    const long plane = (long)H * W;
    const bool on = do_rotate[b] != 0;
    float m0 = 1.f, m1 = 0.f, m3 = 0.f, m4 = 1.f;
    if (on) {
        const double a = (double)angle_deg[b] * 3.14159265358979323846 / 180.0;     // math.radians / math.cos / math.sin (double), then float32 theta
        m0 = (float)cos(a); m1 = (float)(-sin(a)); m3 = (float)sin(a); m4 = (float)cos(a);
    }
    const float t00 = m0 / (0.5f * W), t10 = m1 / (0.5f * W), t01 = m3 / (0.5f * H), t11 = m4 / (0.5f * H);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        const int oy = (int)(i / W), ox = (int)(i - (long)oy * W);
        if (!on) { for (int ch = 0; ch < c; ++ch) dst[((long)b * c + ch) * plane + i] = src[((long)b * c + ch) * plane + i]; continue; }
        const float xc = -0.5f * W + 0.5f + ox, yc = -0.5f * H + 0.5f + oy;
        const float gx = xc * t00 + yc * t10, gy = xc * t01 + yc * t11;
        const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
        if (!bilinear) {
            const float rx = nearbyintf(ix), ry = nearbyintf(iy);
            const bool ok = rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1);
            const long si = ok ? (long)ry * W + (long)rx : 0;
            for (int ch = 0; ch < c; ++ch) dst[((long)b * c + ch) * plane + i] = ok ? src[((long)b * c + ch) * plane + si] : 0.f;
        } else {
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
            const bool okx0 = x0 >= 0 && x0 < W, okx1 = x1 >= 0 && x1 < W, oky0 = y0 >= 0 && y0 < H, oky1 = y1 >= 0 && y1 < H;
            for (int ch = 0; ch < c; ++ch) {
                const float* s = src + ((long)b * c + ch) * plane;
                float v = 0.f;                                   // grid_sample's order: nw, ne, sw, se
                if (oky0 && okx0) v += s[(long)y0 * W + x0] * (wx0 * wy0);
                if (oky0 && okx1) v += s[(long)y0 * W + x1] * (wx1 * wy0);
                if (oky1 && okx0) v += s[(long)y1 * W + x0] * (wx0 * wy1);
                if (oky1 && okx1) v += s[(long)y1 * W + x1] * (wx1 * wy1);
                dst[((long)b * c + ch) * plane + i] = v;
            }
        }
    }
}

// Transforms.resize_and_crop (src/transforms.py:1222-1283): functional.resize(image, (rh, rw), interpolation) = F.interpolate
// (nearest: src = min(floor(dst * in/out), in-1); bilinear, align_corners=False: src = max((dst + 0.5) * in/out - 0.5, 0)), then the
// crop [sy : sy+H, sx : sx+W] back to the original size -- fused: only the cropped window of the resized image is ever computed.
// depth_div: the reference's resize_scaling_depth (image /= rw / W for the non-image tensors).
__global__ void resize_crop_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int H, int W,
                                   const unsigned char* __restrict__ do_it, const int* __restrict__ rh_, const int* __restrict__ rw_,
                                   const int* __restrict__ sy_, const int* __restrict__ sx_, int bilinear, int depth_div) {
    const int b = blockIdx.z;
    const long plane = (long)H * W;
    const bool on = do_it[b] != 0;
    const int rh = rh_[b], rw = rw_[b], sy = sy_[b], sx = sx_[b];
    const float scy = (float)H / (float)rh, scx = (float)W / (float)rw;
    const float div = depth_div ? (float)rw / (float)W : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        if (!on) { for (int ch = 0; ch < c; ++ch) dst[((long)b * c + ch) * plane + i] = src[((long)b * c + ch) * plane + i]; continue; }
        const int oy = (int)(i / W), ox = (int)(i - (long)oy * W);
        const int Y = oy + sy, X = ox + sx;                       // pixel of the resized image
        if (!bilinear) {
            const int yy = min((int)floorf((float)Y * scy), H - 1), xx = min((int)floorf((float)X * scx), W - 1);
            for (int ch = 0; ch < c; ++ch) { const float v = src[((long)b * c + ch) * plane + (long)yy * W + xx]; dst[((long)b * c + ch) * plane + i] = depth_div ? v / div : v; }
        } else {
            float fy = scy * ((float)Y + 0.5f) - 0.5f, fx = scx * ((float)X + 0.5f) - 0.5f;
            fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
            for (int ch = 0; ch < c; ++ch) {
                const float* s = src + ((long)b * c + ch) * plane;
                const float v = ly0 * (lx0 * s[(long)y0 * W + x0] + lx1 * s[(long)y0 * W + x1]) + ly1 * (lx0 * s[(long)y1 * W + x0] + lx1 * s[(long)y1 * W + x1]);
                dst[((long)b * c + ch) * plane + i] = depth_div ? v / div : v;
            }
        }
    }
}

// Photometric jitter on uint8-valued images (Transforms.transform casts float images to uint8 first, src/transforms.py:236-241):
// brightness -> contrast -> saturation, each torchvision's _blend(img1, img2, ratio) = (ratio * img1 + (1 - ratio) * img2).clamp(0, 255)
// truncated back to uint8; img2 = 0 (brightness), the sample's mean of the uint8 grayscale (contrast), the uint8 grayscale (saturation);
// grayscale = trunc(0.2989 r + 0.587 g + 0.114 b).  Contrast needs a per-sample mean of the brightness-adjusted image: two passes.
__device__ __forceinline__ float u8trunc(float v) { v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v); return floorf(v); }
__device__ __forceinline__ float gray_u8(float r, float g, float bl) { return floorf(0.2989f * r + 0.587f * g + 0.114f * bl); }
#define PH_BLOCKS 256
__global__ __launch_bounds__(256) void photo_pass1_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                          const unsigned char* __restrict__ do_b, const float* __restrict__ f_b,
                                                          double* __restrict__ part) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const long plane = (long)H * W;
    const bool on = do_b && do_b[b];
    const float f = on ? f_b[b] : 1.f;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        float v[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float x = u8trunc(src[((long)b * 3 + ch) * plane + i]);                 // .to(torch.uint8)
            if (on) x = u8trunc(f * x);                                              // _blend(img, zeros, f)
            v[ch] = x; dst[((long)b * 3 + ch) * plane + i] = x;
        }
        acc += (double)gray_u8(v[0], v[1], v[2]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)b * PH_BLOCKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void photo_pass2_kernel(float* __restrict__ img, int H, int W, const double* __restrict__ part,
                                                          const unsigned char* __restrict__ do_c, const float* __restrict__ f_c,
                                                          const unsigned char* __restrict__ do_s, const float* __restrict__ f_s) {
    const int b = blockIdx.y;
    const long plane = (long)H * W;
    const bool oc = do_c && do_c[b], os = do_s && do_s[b];
    if (!oc && !os) return;
    float mean = 0.f;
    if (oc) { double s = 0.0; for (int k = 0; k < PH_BLOCKS; ++k) s += part[(long)b * PH_BLOCKS + k]; mean = (float)(s / (double)plane); }
    const float fc = oc ? f_c[b] : 1.f, gc = (float)(1.0 - (double)fc);
    const float fs = os ? f_s[b] : 1.f, gs = (float)(1.0 - (double)fs);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        float v[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { v[ch] = img[((long)b * 3 + ch) * plane + i]; if (oc) v[ch] = u8trunc(fc * v[ch] + gc * mean); }
        if (os) { const float g = gray_u8(v[0], v[1], v[2]);
#pragma unroll
                  for (int ch = 0; ch < 3; ++ch) v[ch] = u8trunc(fs * v[ch] + gs * g); }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) img[((long)b * 3 + ch) * plane + i] = v[ch];
    }
}

extern "C" int ptta_rotate(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_rotate,
                           const float* angle_deg, int bilinear, ptta_stream stream) {
    if (!src || !dst || src == dst || !do_rotate || !angle_deg || n <= 0 || n > 65535 || channels <= 0 || height <= 0 || width <= 0) return -22;
    long blocks = ((long)height * width + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rotate_kernel, dim3((unsigned)blocks, 1, n), dim3(256), 0, (hipStream_t)stream, src, dst, channels, height, width, do_rotate, angle_deg, bilinear ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
extern "C" int ptta_resize_crop(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_resize,
                                const int32_t* resize_height, const int32_t* resize_width, const int32_t* start_y, const int32_t* start_x,
                                int bilinear, int scale_depth, ptta_stream stream) {
    if (!src || !dst || src == dst || !do_resize || !resize_height || !resize_width || !start_y || !start_x || n <= 0 || n > 65535 ||
        channels <= 0 || height <= 0 || width <= 0) return -22;
    long blocks = ((long)height * width + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(resize_crop_kernel, dim3((unsigned)blocks, 1, n), dim3(256), 0, (hipStream_t)stream, src, dst, channels, height, width, do_resize,
                       resize_height, resize_width, start_y, start_x, bilinear ? 1 : 0, scale_depth ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
extern "C" int ptta_photometric(const float* src, float* dst, int n, int height, int width, const uint8_t* do_brightness, const float* f_brightness,
                                const uint8_t* do_contrast, const float* f_contrast, const uint8_t* do_saturation, const float* f_saturation,
                                double* scratch, ptta_stream stream) {
    if (!src || !dst || !scratch || n <= 0 || n > 65535 || height <= 0 || width <= 0) return -22;
    if ((do_brightness && !f_brightness) || (do_contrast && !f_contrast) || (do_saturation && !f_saturation)) return -22;
    hipLaunchKernelGGL(photo_pass1_kernel, dim3(PH_BLOCKS, n), dim3(256), 0, (hipStream_t)stream, src, dst, height, width, do_brightness, f_brightness, scratch);
    if (do_contrast || do_saturation)
        hipLaunchKernelGGL(photo_pass2_kernel, dim3(PH_BLOCKS, n), dim3(256), 0, (hipStream_t)stream, dst, height, width, scratch, do_contrast, f_contrast, do_saturation, f_saturation);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---- gamma, hue, noise, patch removal, crop-and-pad, resize-and-pad: the augmentations of Transforms that no adapt script enables ------
// (src/transforms.py:279-305 gamma / hue between contrast and saturation, :322-332 noise, :508-625 crop-and-pad / resize-and-pad, :630-655
// patch removal).  gamma / hue / pad / resize are torchvision 0.10.1 calls in the reference (parity unpinned, restated); noise, patch removal
// and the crop / pad index arithmetic are the reference's own torch code (pinned: tests/golden/transforms_extra.npz).
// Arithmetic that must round like torch's separate fp32 operations: contraction is switched OFF from here to the end of the file (hipcc fuses
// a * b + c into an fma by default, and HIP's __fmul_rn / __fadd_rn are plain operators that contract all the same: the noise kernel's
// x + spread * z was one ulp off the reference on 1 % of the pixels); the _rn spellings below only mark the operations that matter.
#pragma clang fp contract(off)
__device__ __forceinline__ float rn_add(float a, float b) { return a + b; }       // (defined INSIDE the pragma's scope: the header's __fadd_rn is not)
__device__ __forceinline__ float rn_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float rn_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float aug_remainder1(float a) {            // torch `a % 1.0` for floats: fmod, then the divisor's sign
    float m = fmodf(a, 1.0f);
    if (m != 0.f && m < 0.f) m = rn_add(m, 1.0f);
    return m;
}
__device__ __forceinline__ float aug_clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
// functional_tensor.adjust_hue on one uint8-valued pixel (r, g, b in 0..255): / 255 -> _rgb2hsv -> h = (h + f) % 1 -> _hsv2rgb -> (x * 255) truncated
__device__ __forceinline__ void aug_hue_u8(float& r8, float& g8, float& b8, float hf) {
    const float r = r8 / 255.0f, g = g8 / 255.0f, b = b8 / 255.0f;
    const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
    const bool eqc = maxc == minc;
    const float cr = rn_sub(maxc, minc);
    const float s = cr / (eqc ? 1.0f : maxc);
    const float crd = eqc ? 1.0f : cr;
    const float rc = rn_sub(maxc, r) / crd, gc = rn_sub(maxc, g) / crd, bc = rn_sub(maxc, b) / crd;
    const float hr = (maxc == r) ? rn_sub(bc, gc) : 0.f;
    const float hg = ((maxc == g) && (maxc != r)) ? rn_sub(rn_add(2.0f, rc), bc) : 0.f;
    const float hb = ((maxc != g) && (maxc != r)) ? rn_sub(rn_add(4.0f, gc), rc) : 0.f;
    float h = rn_add(rn_add(hr, hg), hb);
    h = fmodf(rn_add(h / 6.0f, 1.0f), 1.0f);
    h = aug_remainder1(rn_add(h, hf));
    const float v = maxc;
    const float h6 = rn_mul(h, 6.0f);
    const float fi = floorf(h6);
    const float f = rn_sub(h6, fi);
    int i = (int)fi; i = ((i % 6) + 6) % 6;
    const float p = aug_clamp01(rn_mul(v, rn_sub(1.0f, s)));
    const float q = aug_clamp01(rn_mul(v, rn_sub(1.0f, rn_mul(s, f))));
    const float t = aug_clamp01(rn_mul(v, rn_sub(1.0f, rn_mul(s, rn_sub(1.0f, f)))));
    float ro, go, bo;
    switch (i) {
        case 0: ro = v; go = t; bo = p; break;
        case 1: ro = q; go = v; bo = p; break;
        case 2: ro = p; go = v; bo = t; break;
        case 3: ro = p; go = q; bo = v; break;
        case 4: ro = t; go = p; bo = v; break;
        default: ro = v; go = p; bo = q; break;
    }
    r8 = floorf(rn_mul(ro, 255.0f)); g8 = floorf(rn_mul(go, 255.0f)); b8 = floorf(rn_mul(bo, 255.0f));     // .to(uint8) of a value in [0, 255]
}
// functional_tensor.adjust_gamma on a uint8 value: (x / 255) ** gamma clamped to [0, 1], back through convert_image_dtype: floor(x * 255.999)
__device__ __forceinline__ float aug_gamma_u8(float x8, float gamma) {
    const float r = aug_clamp01(powf(x8 / 255.0f, gamma));
    return floorf(rn_mul(r, 255.0f + 1.0f - 1e-3f));
}
// pass 2 with every photometric option: contrast -> gamma -> hue -> saturation on the uint8-valued image pass 1 left in `img`
__global__ __launch_bounds__(256) void photo_pass2_full_kernel(float* __restrict__ img, int H, int W, const double* __restrict__ part,
                                                               const unsigned char* __restrict__ do_c, const float* __restrict__ f_c,
                                                               const unsigned char* __restrict__ do_g, const float* __restrict__ f_g,
                                                               const unsigned char* __restrict__ do_h, const float* __restrict__ f_h,
                                                               const unsigned char* __restrict__ do_s, const float* __restrict__ f_s) {
    const int b = blockIdx.y;
    const long plane = (long)H * W;
    const bool oc = do_c && do_c[b], og = do_g && do_g[b], oh = do_h && do_h[b], os = do_s && do_s[b];
    if (!oc && !og && !oh && !os) return;
    float mean = 0.f;
    if (oc) { double s = 0.0; for (int k = 0; k < PH_BLOCKS; ++k) s += part[(long)b * PH_BLOCKS + k]; mean = (float)(s / (double)plane); }
    const float fc = oc ? f_c[b] : 1.f, gc = (float)(1.0 - (double)fc);
    const float fs = os ? f_s[b] : 1.f, gs = (float)(1.0 - (double)fs);
    const float gam = og ? f_g[b] : 1.f, hf = oh ? f_h[b] : 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        float v[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { v[ch] = img[((long)b * 3 + ch) * plane + i]; if (oc) v[ch] = u8trunc(fc * v[ch] + gc * mean); }
        if (og) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) v[ch] = aug_gamma_u8(v[ch], gam);
        }
        if (oh) aug_hue_u8(v[0], v[1], v[2], hf);
        if (os) { const float g = gray_u8(v[0], v[1], v[2]);
#pragma unroll
                  for (int ch = 0; ch < 3; ++ch) v[ch] = u8trunc(fs * v[ch] + gs * g); }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) img[((long)b * 3 + ch) * plane + i] = v[ch];
    }
}
// gamma as the ONLY photometric option: the reference leaves the images float (do_photometric_transforms does not count gamma, :102-106)
// and torchvision's float branch is (x ** gamma).clamp(0, 1) on the values as they are
__global__ void gamma_float_kernel(const float* __restrict__ src, float* __restrict__ dst, long per, const unsigned char* __restrict__ do_g,
                                   const float* __restrict__ f_g) {
    const int b = blockIdx.y;
    const bool on = do_g[b] != 0;
    const float g = f_g[b];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        const float x = src[(long)b * per + i];
        dst[(long)b * per + i] = on ? aug_clamp01(powf(x, g)) : x;
    }
}
extern "C" int ptta_photometric_full(const float* src, float* dst, int n, int height, int width, const uint8_t* do_brightness, const float* f_brightness,
                                     const uint8_t* do_contrast, const float* f_contrast, const uint8_t* do_gamma, const float* f_gamma,
                                     const uint8_t* do_hue, const float* f_hue, const uint8_t* do_saturation, const float* f_saturation,
                                     double* scratch, ptta_stream stream) {
    if (!src || !dst || src == dst || n <= 0 || n > 65535 || height <= 0 || width <= 0) return -22;
    if ((do_brightness && !f_brightness) || (do_contrast && !f_contrast) || (do_saturation && !f_saturation) || (do_gamma && !f_gamma) || (do_hue && !f_hue)) return -22;
    const bool as_u8 = do_brightness || do_contrast || do_hue || do_saturation;
    if (!as_u8) {
        if (!do_gamma) return -22;
        const long per = 3L * height * width;
        long blocks = (per + 255) / 256; if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gamma_float_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, src, dst, per, do_gamma, f_gamma);
        return hipGetLastError() == hipSuccess ? 0 : -5;
    }
    if (!scratch) return -22;
    hipLaunchKernelGGL(photo_pass1_kernel, dim3(PH_BLOCKS, n), dim3(256), 0, (hipStream_t)stream, src, dst, height, width, do_brightness, f_brightness, scratch);
    if (do_contrast || do_gamma || do_hue || do_saturation)
        hipLaunchKernelGGL(photo_pass2_full_kernel, dim3(PH_BLOCKS, n), dim3(256), 0, (hipStream_t)stream, dst, height, width, scratch, do_contrast, f_contrast,
                           do_gamma, f_gamma, do_hue, f_hue, do_saturation, f_saturation);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// Transforms.add_noise (:839-876): image + spread * noise (gaussian) or image + spread * (noise - 0.5) (uniform) for the samples whose coin came
// up; `noise` = the caller's torch.randn / torch.rand field (the random stream stays with the caller, like every other draw)
__global__ void add_noise_kernel(const float* __restrict__ src, const float* __restrict__ noise, float* __restrict__ dst, long per,
                                 const unsigned char* __restrict__ do_it, float spread, int uniform) {
    const int b = blockIdx.y;
    const bool on = do_it[b] != 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        const float x = src[(long)b * per + i];
        float z = on ? noise[(long)b * per + i] : 0.f;
        if (uniform) z = rn_sub(z, 0.5f);
        dst[(long)b * per + i] = on ? rn_add(x, rn_mul(spread, z)) : x;
    }
}
extern "C" int ptta_add_noise(const float* src, const float* noise, float* dst, int n, int channels, int height, int width, const uint8_t* do_noise,
                              float spread, int uniform, ptta_stream stream) {
    if (!src || !noise || !dst || !do_noise || n <= 0 || n > 65535 || channels <= 0 || height <= 0 || width <= 0) return -22;
    const long per = (long)channels * height * width;
    long blocks = (per + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, src, noise, dst, per, do_noise, spread, uniform ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// Transforms.remove_random_patches (:878-924) behind random_nonzero's selection (`selected`: n x H x W bytes, 1 at the chosen nonzero pixels):
// mask = (sum_c |image| > 0); the chosen pixels become +inf; max_pool2d(kernel (ph, pw), stride 1, padding (ph / 2, pw / 2)) spreads them over
// their patch; inf -> 0; image * mask.  Output pixel (y, x) is removed when a chosen pixel lies in rows y - ph/2 .. y - ph/2 + ph - 1 and
// columns x - pw/2 .. + pw - 1 (odd sizes: the patch centred on it); every other pixel is mask * image = itself (an all-zero pixel has mask 0).
__global__ void remove_patches_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int H, int W,
                                      const unsigned char* __restrict__ do_it, const unsigned char* __restrict__ selected,
                                      const int* __restrict__ ph_, const int* __restrict__ pw_) {
    const int b = blockIdx.z;
    const long plane = (long)H * W;
    const bool on = do_it[b] != 0;
    const int ph = ph_[b], pw = pw_[b];
    const unsigned char* sel = selected + (long)b * plane;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        bool removed = false;
        if (on) {
            const int y = (int)(i / W), x = (int)(i - (long)y * W);
            const int y0 = max(y - ph / 2, 0), y1 = min(y - ph / 2 + ph - 1, H - 1), x0 = max(x - pw / 2, 0), x1 = min(x - pw / 2 + pw - 1, W - 1);
            for (int yy = y0; yy <= y1 && !removed; ++yy)
                for (int xx = x0; xx <= x1; ++xx) if (sel[(long)yy * W + xx]) { removed = true; break; }
        }
        for (int ch = 0; ch < c; ++ch) { const float v = src[((long)b * c + ch) * plane + i]; dst[((long)b * c + ch) * plane + i] = removed ? 0.f : v; }
    }
}
extern "C" int ptta_remove_patches(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_remove,
                                   const uint8_t* selected, const int32_t* patch_height, const int32_t* patch_width, ptta_stream stream) {
    if (!src || !dst || src == dst || !do_remove || !selected || !patch_height || !patch_width || n <= 0 || n > 65535 || channels <= 0 || height <= 0 || width <= 0) return -22;
    long blocks = ((long)height * width + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(remove_patches_kernel, dim3((unsigned)blocks, 1, n), dim3(256), 0, (hipStream_t)stream, src, dst, channels, height, width, do_remove, selected,
                       patch_height, patch_width);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// index of a padded coordinate q (relative to an image of `size` pixels) under torchvision's padding modes: 1 edge, 2 reflect (no edge
// repeat), 3 symmetric (edge repeated); returns -1 for constant padding
__device__ __forceinline__ int aug_pad_index(int q, int size, int mode) {
    if (q >= 0 && q < size) return q;
    if (mode == 0) return -1;
    if (mode == 1) return min(max(q, 0), size - 1);
    if (mode == 2) { int r = q < 0 ? -q : 2 * (size - 1) - q; return min(max(r, 0), size - 1); }
    int r = q < 0 ? -q - 1 : 2 * size - 1 - q;
    return min(max(r, 0), size - 1);
}
// Transforms.crop_and_pad (:1072-1135): image[sy:ey, sx:ex] padded by (left, top, right, bottom) back to H x W
__global__ void crop_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int H, int W, const unsigned char* __restrict__ do_it,
                                const int* __restrict__ sy_, const int* __restrict__ sx_, const int* __restrict__ ey_, const int* __restrict__ ex_,
                                const int* __restrict__ pt_, const int* __restrict__ pl_, int mode, float fill) {
    const int b = blockIdx.z;
    const long plane = (long)H * W;
    const bool on = do_it[b] != 0;
    const int sy = sy_[b], sx = sx_[b], eh = ey_[b] - sy, ew = ex_[b] - sx, pt = pt_[b], pl = pl_[b];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        long si = i;
        bool inside = true;
        if (on) {
            const int y = (int)(i / W), x = (int)(i - (long)y * W);
            const int cy = aug_pad_index(y - pt, eh, mode), cx = aug_pad_index(x - pl, ew, mode);
            inside = cy >= 0 && cx >= 0;
            si = inside ? (long)(sy + cy) * W + (sx + cx) : 0;
        }
        for (int ch = 0; ch < c; ++ch) dst[((long)b * c + ch) * plane + i] = inside ? src[((long)b * c + ch) * plane + si] : fill;
    }
}
extern "C" int ptta_crop_pad(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_crop_pad,
                             const int32_t* start_y, const int32_t* start_x, const int32_t* end_y, const int32_t* end_x,
                             const int32_t* pad_top, const int32_t* pad_left, int padding_mode, float fill, ptta_stream stream) {
    if (!src || !dst || src == dst || !do_crop_pad || !start_y || !start_x || !end_y || !end_x || !pad_top || !pad_left || n <= 0 || n > 65535 ||
        channels <= 0 || height <= 0 || width <= 0 || padding_mode < 0 || padding_mode > 3) return -22;
    long blocks = ((long)height * width + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(crop_pad_kernel, dim3((unsigned)blocks, 1, n), dim3(256), 0, (hipStream_t)stream, src, dst, channels, height, width, do_crop_pad,
                       start_y, start_x, end_y, end_x, pad_top, pad_left, padding_mode, fill);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
// Transforms.resize_and_pad (:1137-1220): functional.resize to (rh, rw) <= (H, W) (the sampling of resize_crop_kernel), padded back to H x W
__global__ void resize_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int H, int W, const unsigned char* __restrict__ do_it,
                                  const int* __restrict__ rh_, const int* __restrict__ rw_, const int* __restrict__ pt_, const int* __restrict__ pl_,
                                  int bilinear, int mode, float fill) {
    const int b = blockIdx.z;
    const long plane = (long)H * W;
    const bool on = do_it[b] != 0;
    const int rh = rh_[b], rw = rw_[b], pt = pt_[b], pl = pl_[b];
    const float scy = (float)H / (float)rh, scx = (float)W / (float)rw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        if (!on) { for (int ch = 0; ch < c; ++ch) dst[((long)b * c + ch) * plane + i] = src[((long)b * c + ch) * plane + i]; continue; }
        const int oy = (int)(i / W), ox = (int)(i - (long)oy * W);
        const int Y = aug_pad_index(oy - pt, rh, mode), X = aug_pad_index(ox - pl, rw, mode);      // pixel of the resized image
        if (Y < 0 || X < 0) { for (int ch = 0; ch < c; ++ch) dst[((long)b * c + ch) * plane + i] = fill; continue; }
        if (!bilinear) {
            const int yy = min((int)floorf((float)Y * scy), H - 1), xx = min((int)floorf((float)X * scx), W - 1);
            for (int ch = 0; ch < c; ++ch) dst[((long)b * c + ch) * plane + i] = src[((long)b * c + ch) * plane + (long)yy * W + xx];
        } else {
            float fy = scy * ((float)Y + 0.5f) - 0.5f, fx = scx * ((float)X + 0.5f) - 0.5f;
            fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
            for (int ch = 0; ch < c; ++ch) {
                const float* s = src + ((long)b * c + ch) * plane;
                dst[((long)b * c + ch) * plane + i] = ly0 * (lx0 * s[(long)y0 * W + x0] + lx1 * s[(long)y0 * W + x1]) + ly1 * (lx0 * s[(long)y1 * W + x0] + lx1 * s[(long)y1 * W + x1]);
            }
        }
    }
}
extern "C" int ptta_resize_pad(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_resize_pad,
                               const int32_t* resize_height, const int32_t* resize_width, const int32_t* pad_top, const int32_t* pad_left,
                               int bilinear, int padding_mode, float fill, ptta_stream stream) {
    if (!src || !dst || src == dst || !do_resize_pad || !resize_height || !resize_width || !pad_top || !pad_left || n <= 0 || n > 65535 ||
        channels <= 0 || height <= 0 || width <= 0 || padding_mode < 0 || padding_mode > 3) return -22;
    long blocks = ((long)height * width + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(resize_pad_kernel, dim3((unsigned)blocks, 1, n), dim3(256), 0, (hipStream_t)stream, src, dst, channels, height, width, do_resize_pad,
                       resize_height, resize_width, pad_top, pad_left, bilinear ? 1 : 0, padding_mode, fill);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
