// Input side: temporal sub-sampling + anti-aliased bilinear resize + per-channel normalisation of decoded
// uint8 frames on the GPU — utils/video.py:246-288 of the reference does this on the host in fp32 (frame index select by
// torch.linspace(0, t-1, n).long() / last-frame padding, torchvision v2.Resize((s, s), antialias=True) = F.interpolate(
// mode="bilinear", antialias=True, align_corners=False), v2.Normalize).  The clip crosses PCIe as raw bytes at its native
// size and leaves this kernel as the NDHWC fp32 tensor (or re-quantised uint8 for dcc_im2col3d_u8) the trunk reads.
//
// Arithmetic follows ATen's separable anti-aliased kernel (UpSampleKernel.cpp, _compute_indices_min_size_weights_aa +
// HelperInterpLinear::aa_filter) in fp32: per output coordinate i along an axis of length `in` resized to `out`
//   scale = in / out;  support = scale >= 1 ? scale : 1;  center = scale * (i + 0.5)
//   xmin = max(int(center - support + 0.5), 0);  xsize = min(int(center + support + 0.5), in) - xmin
//   w[j] = triangle((j + xmin - center + 0.5) * (scale >= 1 ? 1 / scale : 1)),  normalised by their sum
// horizontal pass first, vertical second (ATen interpolates the last dimension first), both in fp32.
// HBM-bound byte work: one thread per output pixel, 3 channels; a 512^2 -> 224^2 clip reads 12.6 MB and writes 9.6 MB.
#include "common.h"

namespace {

struct ResizeGeom {
  int T_out, H_in, W_in, C_in, H_out, W_out;
  float sy, sx;       // in / out per axis
  int normalize;      // apply (v - mean[c]) / std[c]
  int out_u8;         // 1: round-half-even + clamp to [0, 255], uint8 output
  int antialias;
};

__device__ __forceinline__ float tri(float x) {
  x = fabsf(x);
  return x < 1.0f ? 1.0f - x : 0.0f;
}

// window of output coordinate i: first input index, tap count, 1 / sum of the raw weights; w(j) = tri((j + xmin - center + 0.5) * inv)
__device__ __forceinline__ void window(int i, int in, float scale, int antialias, int& xmin, int& xsize, float& center, float& inv,
                                       float& rnorm) {
  const bool down = antialias && scale >= 1.0f;
  const float support = down ? scale : 1.0f;
  inv = down ? 1.0f / scale : 1.0f;
  center = scale * ((float)i + 0.5f);
  xmin = max((int)(center - support + 0.5f), 0);
  xsize = min((int)(center + support + 0.5f), in) - xmin;
  float tot = 0.0f;
  for (int j = 0; j < xsize; ++j) tot += tri(((float)(j + xmin) - center + 0.5f) * inv);
  rnorm = tot != 0.0f ? 1.0f / tot : 0.0f;
}

__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, const int* __restrict__ frame_idx,
                                                        void* __restrict__ dst, const float* __restrict__ mean,
                                                        const float* __restrict__ stdv, ResizeGeom g) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int t = blockIdx.z;
  if (ox >= g.W_out || oy >= g.H_out) return;
  const int ft = frame_idx ? frame_idx[t] : t;
  const uint8_t* frame = src + (int64_t)ft * g.H_in * g.W_in * g.C_in;
  int x0, xn, y0, yn;
  float cx, ix, nx, cy, iy, ny;
  window(ox, g.W_in, g.sx, g.antialias, x0, xn, cx, ix, nx);
  window(oy, g.H_in, g.sy, g.antialias, y0, yn, cy, iy, ny);
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int jy = 0; jy < yn; ++jy) {
    const float wy = tri(((float)(jy + y0) - cy + 0.5f) * iy) * ny;
    const uint8_t* row = frame + ((int64_t)(y0 + jy) * g.W_in + x0) * g.C_in;
    float h[3] = {0.0f, 0.0f, 0.0f};   // horizontal pass of this input row (ATen: last dimension first)
    for (int jx = 0; jx < xn; ++jx) {
      const float wx = tri(((float)(jx + x0) - cx + 0.5f) * ix) * nx;
      if (g.C_in == 3) {
        h[0] += wx * (float)row[jx * 3 + 0];
        h[1] += wx * (float)row[jx * 3 + 1];
        h[2] += wx * (float)row[jx * 3 + 2];
      } else {
        h[0] += wx * (float)row[jx];
      }
    }
    acc[0] += wy * h[0];
    acc[1] += wy * h[1];
    acc[2] += wy * h[2];
  }
  if (g.C_in == 1) acc[1] = acc[2] = acc[0];   // grayscale -> 3 identical channels (utils/video.py:215-221)
  const int64_t o = (((int64_t)t * g.H_out + oy) * g.W_out + ox) * 3;
  if (g.out_u8) {
    uint8_t* d = reinterpret_cast<uint8_t*>(dst) + o;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = (uint8_t)fminf(fmaxf(rintf(acc[c]), 0.0f), 255.0f);
  } else {
    float* d = reinterpret_cast<float*>(dst) + o;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = g.normalize ? (acc[c] - mean[c]) / stdv[c] : acc[c];
  }
}

}  // namespace

// src: uint8 [T_in, H_in, W_in, C_in] (C_in 1 or 3); frame_idx: device int32 [T_out] source frame of every output frame
// (null: identity, needs T_out <= T_in); dst: fp32 (out_u8 = 0) or uint8 (out_u8 = 1) [T_out, H_out, W_out, 3];
// mean / stdv: device fp32 [3] or null (no normalisation; ignored for uint8 output).
extern "C" int dcc_resize_frames_u8(const uint8_t* src, const int* frame_idx, void* dst, const float* mean, const float* stdv, int T_in,
                                    int H_in, int W_in, int C_in, int T_out, int H_out, int W_out, int antialias, int out_u8,
                                    void* stream) {
  if (!src || !dst || T_in <= 0 || H_in <= 0 || W_in <= 0 || T_out <= 0 || H_out <= 0 || W_out <= 0) return DCC_ERR_ARG;
  if ((C_in != 1 && C_in != 3) || (!frame_idx && T_out > T_in) || ((mean == nullptr) != (stdv == nullptr))) return DCC_ERR_ARG;
  if (T_out > 65535 || (H_out + 3) / 4 > 65535) return DCC_ERR_UNSUPPORTED;
  ResizeGeom g{T_out, H_in, W_in, C_in, H_out, W_out, (float)H_in / (float)H_out, (float)W_in / (float)W_out,
               mean != nullptr && !out_u8, out_u8 != 0, antialias != 0};
  const dim3 grid((W_out + 63) / 64, (H_out + 3) / 4, T_out);
  hipLaunchKernelGGL(resize_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, frame_idx, dst, mean, stdv, g);
  return dcc_check_launch();
}
