// Label painting after predict()  (reference Annotator.colorize, cell_type_annotation/model.py:806-858, n_regions = 0):
// every pixel of a cell gets the colour of the cell's type, the colour of its confidence, and its type index + 1; background 0.
// The reference loops over cells and fancy-indexes their pixel lists (seconds to minutes at 1e5 cells); here it is one gather per
// pixel through a label -> cell table.  HBM-bound: 4 bytes read + 7 bytes written per pixel.
#include <algorithm>

#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

__global__ __launch_bounds__(256) void colorize_kernel(const int32_t* __restrict__ mask, long long npx, const int32_t* __restrict__ label_to_cell,
                                                       int L, const uint8_t* __restrict__ type_rgb, const uint8_t* __restrict__ conf_rgb,
                                                       const uint8_t* __restrict__ type_idx, uint8_t* __restrict__ out_type,
                                                       uint8_t* __restrict__ out_conf, uint8_t* __restrict__ out_idx) {
  const long long groups = (npx + 3) >> 2;
  for (long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups; gidx += (long long)gridDim.x * blockDim.x) {
    const long long p0 = gidx << 2;
    uint8_t a[12], b[12], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int cell = -1;
      if (p0 + i < npx) {
        const int lab = mask[p0 + i];
        if (lab > 0 && lab < L) cell = label_to_cell[lab];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        a[3 * i + k] = cell >= 0 ? type_rgb[3 * (size_t)cell + k] : 0;
        b[3 * i + k] = cell >= 0 ? conf_rgb[3 * (size_t)cell + k] : 0;
      }
      c[i] = cell >= 0 ? type_idx[cell] : 0;
    }
    if (p0 + 3 < npx) {          // 12-byte and 4-byte groups start on 4-byte boundaries
      uint32_t* oa = reinterpret_cast<uint32_t*>(out_type + 3 * p0);
      uint32_t* ob = reinterpret_cast<uint32_t*>(out_conf + 3 * p0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        oa[k] = a[4 * k] | (a[4 * k + 1] << 8) | (a[4 * k + 2] << 16) | ((uint32_t)a[4 * k + 3] << 24);
        ob[k] = b[4 * k] | (b[4 * k + 1] << 8) | (b[4 * k + 2] << 16) | ((uint32_t)b[4 * k + 3] << 24);
      }
      *reinterpret_cast<uint32_t*>(out_idx + p0) = c[0] | (c[1] << 8) | (c[2] << 16) | ((uint32_t)c[3] << 24);
    } else {
      for (int i = 0; p0 + i < npx; ++i) {
        for (int k = 0; k < 3; ++k) {
          out_type[3 * (p0 + i) + k] = a[3 * i + k];
          out_conf[3 * (p0 + i) + k] = b[3 * i + k];
        }
        out_idx[p0 + i] = c[i];
      }
    }
  }
}

void launch_colorize(const int32_t* mask, long long npx, const int32_t* label_to_cell, int L, const uint8_t* type_rgb, const uint8_t* conf_rgb,
                     const uint8_t* type_idx, uint8_t* out_type, uint8_t* out_conf, uint8_t* out_idx, hipStream_t s) {
  if (npx <= 0) return;
  const long long groups = (npx + 3) >> 2;
  const int blocks = (int)std::min<long long>((groups + 255) / 256, 16384);
  hipLaunchKernelGGL(colorize_kernel, dim3(blocks), dim3(256), 0, s, mask, npx, label_to_cell, L, type_rgb, conf_rgb, type_idx, out_type, out_conf,
                     out_idx);
}

}  // namespace ribca
