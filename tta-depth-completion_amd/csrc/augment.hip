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
