// K1: point cloud -> BEV occupancy, sorted-unique index list, and the batched
// dense rebuild.  HBM-bound integer/byte work: one coalesced pass over the
// points, idempotent plain stores into the occupancy grid (no atomics needed:
// every writer stores the same 1.0f), float64 divide so floor() is bit-exact
// with numpy (SURVEY.md Appx A.2).
//
// Replaces upstream:coperception/utils/data_util.py :: voxelize_occupy and the
// dense rebuild of upstream:coperception/datasets/V2XSimDet.py :: __getitem__
// (SURVEY.md §8 a1, a2).
#include "dn_internal.h"

namespace {

struct VoxelGeom {
  double vx, vy, vz;
  double xlo, xhi, ylo, yhi, zlo, zhi;
  int minx, miny, minz;
  int dx, dy, dz;
};

__global__ void voxelize_kernel(const float* __restrict__ pts, int n, int stride, VoxelGeom g,
                                float* __restrict__ dense) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = pts + (size_t)i * stride;
    // numpy compares the float32 coordinates against float64 extents (strict <)
    const double x = (double)p[0], y = (double)p[1], z = (double)p[2];
    if (!(g.xlo < x && x < g.xhi && g.ylo < y && y < g.yhi && g.zlo < z && z < g.zhi)) continue;
    const int qx = (int)floor(x / g.vx) - g.minx;
    const int qy = (int)floor(y / g.vy) - g.miny;
    const int qz = (int)floor(z / g.vz) - g.minz;
    if ((unsigned)qx < (unsigned)g.dx && (unsigned)qy < (unsigned)g.dy && (unsigned)qz < (unsigned)g.dz)
      dense[((size_t)qx * g.dy + qy) * g.dz + qz] = 1.0f;
  }
}

constexpr int COMPACT_BLOCK = 256;
constexpr int COMPACT_ITEMS = 4;
constexpr int COMPACT_TILE = COMPACT_BLOCK * COMPACT_ITEMS;

__device__ inline int block_exclusive_scan(int v, int* lds, int* total) {
  // 256-thread exclusive scan: wave-level shuffles then a 4-entry combine
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += lds[w];
  if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + inc - v;
}

__global__ void compact_count_kernel(const float* __restrict__ dense, long ncell,
                                     int* __restrict__ tile_counts) {
  __shared__ int lds[4];
  const long base = (long)blockIdx.x * COMPACT_TILE + threadIdx.x * COMPACT_ITEMS;
  int c = 0;
#pragma unroll
  for (int e = 0; e < COMPACT_ITEMS; ++e)
    if (base + e < ncell && dense[base + e] != 0.f) ++c;
  int total;
  block_exclusive_scan(c, lds, &total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

__global__ void compact_scan_kernel(int* __restrict__ tile_counts, int ntiles,
                                    int32_t* __restrict__ count_out) {
  // single workgroup: exclusive scan of the per-tile counts, in place
  __shared__ int lds[4];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int start = 0; start < ntiles; start += COMPACT_BLOCK) {
    const int i = start + threadIdx.x;
    const int v = i < ntiles ? tile_counts[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, lds, &total);
    if (i < ntiles) tile_counts[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count_out = carry;
}

__global__ void compact_write_kernel(const float* __restrict__ dense, long ncell, int dy, int dz,
                                     const int* __restrict__ tile_offsets, int capacity,
                                     int32_t* __restrict__ indices) {
  __shared__ int lds[4];
  const long base = (long)blockIdx.x * COMPACT_TILE + threadIdx.x * COMPACT_ITEMS;
  bool occ[COMPACT_ITEMS];
  int c = 0;
#pragma unroll
  for (int e = 0; e < COMPACT_ITEMS; ++e) {
    occ[e] = base + e < ncell && dense[base + e] != 0.f;
    c += occ[e];
  }
  int pos = tile_offsets[blockIdx.x] + block_exclusive_scan(c, lds, nullptr);
#pragma unroll
  for (int e = 0; e < COMPACT_ITEMS; ++e) {
    if (occ[e]) {
      if (pos < capacity) {
        const long cell = base + e;
        const int iz = (int)(cell % dz);
        const long r = cell / dz;
        indices[3 * (long)pos + 0] = (int32_t)(r / dy);
        indices[3 * (long)pos + 1] = (int32_t)(r % dy);
        indices[3 * (long)pos + 2] = iz;
      }
      ++pos;
    }
  }
}

__global__ void scatter_dense_kernel(const int32_t* __restrict__ indices,
                                     const int32_t* __restrict__ offsets, int n_images, int total,
                                     int dx, int dy, int dz, float* __restrict__ dense) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    // image that owns row i: last g with offsets[g] <= i
    int lo = 0, hi = n_images;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const int ix = indices[3 * (size_t)i], iy = indices[3 * (size_t)i + 1],
              iz = indices[3 * (size_t)i + 2];
    if ((unsigned)ix < (unsigned)dx && (unsigned)iy < (unsigned)dy && (unsigned)iz < (unsigned)dz)
      dense[(((size_t)lo * dx + ix) * dy + iy) * dz + iz] = 1.0f;
  }
}

// the same rebuild, straight into the split-planar layout of the conv engine (sp_layout.h):
// occupancy 1.0 is the f16 0x3C00 in the hi plane of the voxel's z bin, its lo plane stays 0
__global__ void scatter_dense_sp_kernel(const int32_t* __restrict__ indices,
                                        const int32_t* __restrict__ offsets, int n_images, int total,
                                        int dx, int dy, int dz, int quarters, unsigned short* __restrict__ sp) {
  const size_t hw = (size_t)dx * dy;
  const int chunks = (dz + 15) / 16;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int lo = 0, hi = n_images;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const int ix = indices[3 * (size_t)i], iy = indices[3 * (size_t)i + 1],
              iz = indices[3 * (size_t)i + 2];
    if ((unsigned)ix < (unsigned)dx && (unsigned)iy < (unsigned)dy && (unsigned)iz < (unsigned)dz) {
      const int cg = iz >> 4, oct = (iz >> 3) & 1, e = iz & 7;
      sp[((((size_t)lo * chunks + cg) * quarters + oct) * hw + (size_t)ix * dy + iy) * 8 + e] = 0x3C00;
    }
  }
}

// the same rebuild as ONE occupancy word per pixel (bit z = bin z holds a point; dz <= 32): 1/32 of the float32 grid.
// Bins of one pixel share a word, so the store is an atomic OR (still idempotent: duplicates and the order of the
// rows cannot change the result).  conv_sp.hip expands the words into f16 0/1 fragments in LDS (math = 4).
__global__ void scatter_dense_bits_kernel(const int32_t* __restrict__ indices,
                                          const int32_t* __restrict__ offsets, int n_images, int total,
                                          int dx, int dy, int dz, unsigned* __restrict__ bits) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int lo = 0, hi = n_images;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const int ix = indices[3 * (size_t)i], iy = indices[3 * (size_t)i + 1],
              iz = indices[3 * (size_t)i + 2];
    if ((unsigned)ix < (unsigned)dx && (unsigned)iy < (unsigned)dy && (unsigned)iz < (unsigned)dz)
      atomicOr(&bits[((size_t)lo * dx + ix) * dy + iy], 1u << iz);
  }
}

inline int ntiles_of(const int* dims) {
  const long ncell = (long)dims[0] * dims[1] * dims[2];
  return (int)((ncell + COMPACT_TILE - 1) / COMPACT_TILE);
}

}  // namespace

extern "C" int dn_voxelize_occupy(const float* pts, int n_pts, int pt_stride,
                                  const double* vs, const double* ext, const int* dims,
                                  float* dense, void* stream) {
  DN_REQUIRE(vs && ext && dims && dense, "voxelize: null pointer");
  DN_REQUIRE(n_pts >= 0 && (n_pts == 0 || pts), "voxelize: bad point buffer");
  DN_REQUIRE(pt_stride >= 3, "voxelize: points need >= 3 columns (got %d)", pt_stride);
  DN_REQUIRE(vs[0] > 0 && vs[1] > 0 && vs[2] > 0, "voxelize: voxel size must be positive");
  VoxelGeom g;
  g.vx = vs[0]; g.vy = vs[1]; g.vz = vs[2];
  g.xlo = ext[0]; g.xhi = ext[1]; g.ylo = ext[2]; g.yhi = ext[3]; g.zlo = ext[4]; g.zhi = ext[5];
  // min_voxel_coord = floor(extent_lo / voxel); dims = ceil(extent_hi / voxel) - 1 - min + 1
  g.minx = (int)floor(ext[0] / vs[0]); g.miny = (int)floor(ext[2] / vs[1]);
  g.minz = (int)floor(ext[4] / vs[2]);
  const int ex = (int)ceil(ext[1] / vs[0]) - g.minx, ey = (int)ceil(ext[3] / vs[1]) - g.miny,
            ez = (int)ceil(ext[5] / vs[2]) - g.minz;
  DN_REQUIRE(dims[0] == ex && dims[1] == ey && dims[2] == ez,
             "voxelize: dims (%d,%d,%d) do not match extents/voxel_size (%d,%d,%d)", dims[0],
             dims[1], dims[2], ex, ey, ez);
  g.dx = dims[0]; g.dy = dims[1]; g.dz = dims[2];
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = (size_t)g.dx * g.dy * g.dz * sizeof(float);
  hipError_t e = dn::zero_fill(dense, bytes, s);
  if (e != hipSuccess) return dn::fail(DN_ERR_LAUNCH, "voxelize: memset: %s", hipGetErrorString(e));
  if (n_pts == 0) return DN_OK;
  const int blocks = (n_pts + 255) / 256 < 2048 ? (n_pts + 255) / 256 : 2048;
  hipLaunchKernelGGL(voxelize_kernel, dim3(blocks), dim3(256), 0, s, pts, n_pts, pt_stride, g, dense);
  return dn::check_launch("voxelize_kernel");
}

extern "C" size_t dn_voxel_compact_workspace(const int* dims) {
  if (!dims) return 0;
  return (size_t)ntiles_of(dims) * sizeof(int);
}

extern "C" int dn_voxel_compact(const float* dense, const int* dims, int32_t* indices,
                                int capacity, int32_t* count, void* workspace, void* stream) {
  DN_REQUIRE(dense && dims && count && workspace, "voxel_compact: null pointer");
  DN_REQUIRE(capacity >= 0 && (capacity == 0 || indices), "voxel_compact: bad index buffer");
  const long ncell = (long)dims[0] * dims[1] * dims[2];
  DN_REQUIRE(ncell > 0, "voxel_compact: empty grid");
  const int ntiles = ntiles_of(dims);
  hipStream_t s = (hipStream_t)stream;
  int* tiles = static_cast<int*>(workspace);
  hipLaunchKernelGGL(compact_count_kernel, dim3(ntiles), dim3(COMPACT_BLOCK), 0, s, dense, ncell, tiles);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(COMPACT_BLOCK), 0, s, tiles, ntiles, count);
  hipLaunchKernelGGL(compact_write_kernel, dim3(ntiles), dim3(COMPACT_BLOCK), 0, s, dense, ncell,
                     dims[1], dims[2], tiles, capacity, indices);
  return dn::check_launch("voxel_compact");
}

extern "C" int dn_scatter_dense(const int32_t* indices, const int32_t* offsets, int n_images,
                                int total, const int* dims, float* dense, void* stream) {
  DN_REQUIRE(dims && dense && offsets, "scatter_dense: null pointer");
  DN_REQUIRE(n_images > 0 && total >= 0 && (total == 0 || indices), "scatter_dense: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = (size_t)n_images * dims[0] * dims[1] * dims[2] * sizeof(float);
  hipError_t e = dn::zero_fill(dense, bytes, s);
  if (e != hipSuccess) return dn::fail(DN_ERR_LAUNCH, "scatter_dense: memset: %s", hipGetErrorString(e));
  if (total == 0) return DN_OK;
  const int blocks = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
  hipLaunchKernelGGL(scatter_dense_kernel, dim3(blocks), dim3(256), 0, s, indices, offsets,
                     n_images, total, dims[0], dims[1], dims[2], dense);
  return dn::check_launch("scatter_dense_kernel");
}

namespace {
int scatter_dense_sp_impl(const int32_t* indices, const int32_t* offsets, int n_images, int total, const int* dims,
                          void* dense_sp, int quarters, hipStream_t s) {
  DN_REQUIRE(dims && dense_sp && offsets, "scatter_dense_sp: null pointer");
  DN_REQUIRE(n_images > 0 && total >= 0 && (total == 0 || indices), "scatter_dense_sp: bad sizes");
  const size_t bytes = (size_t)n_images * ((dims[2] + 15) / 16) * quarters * dims[0] * dims[1] * 16;
  hipError_t e = dn::zero_fill(dense_sp, bytes, s);
  if (e != hipSuccess) return dn::fail(DN_ERR_LAUNCH, "scatter_dense_sp: memset: %s", hipGetErrorString(e));
  if (total == 0) return DN_OK;
  const int blocks = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
  hipLaunchKernelGGL(scatter_dense_sp_kernel, dim3(blocks), dim3(256), 0, s, indices, offsets, n_images,
                     total, dims[0], dims[1], dims[2], quarters, (unsigned short*)dense_sp);
  return dn::check_launch("scatter_dense_sp_kernel");
}
}  // namespace

extern "C" int dn_scatter_dense_sp(const int32_t* indices, const int32_t* offsets, int n_images,
                                   int total, const int* dims, void* dense_sp, void* stream) {
  return scatter_dense_sp_impl(indices, offsets, n_images, total, dims, dense_sp, 4, (hipStream_t)stream);
}

extern "C" int dn_scatter_dense_sp_hi(const int32_t* indices, const int32_t* offsets, int n_images,
                                      int total, const int* dims, void* dense_sp_hi, void* stream) {
  return scatter_dense_sp_impl(indices, offsets, n_images, total, dims, dense_sp_hi, 2, (hipStream_t)stream);
}

extern "C" int dn_scatter_dense_bits(const int32_t* indices, const int32_t* offsets, int n_images,
                                     int total, const int* dims, uint32_t* bits, void* stream) {
  DN_REQUIRE(dims && bits && offsets, "scatter_dense_bits: null pointer");
  DN_REQUIRE(n_images > 0 && total >= 0 && (total == 0 || indices), "scatter_dense_bits: bad sizes");
  DN_REQUIRE(dims[2] >= 1 && dims[2] <= 32, "scatter_dense_bits: %d height bins do not fit one word per pixel", dims[2]);
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = (size_t)n_images * dims[0] * dims[1] * sizeof(uint32_t);
  hipError_t e = dn::zero_fill(bits, bytes, s);
  if (e != hipSuccess) return dn::fail(DN_ERR_LAUNCH, "scatter_dense_bits: memset: %s", hipGetErrorString(e));
  if (total == 0) return DN_OK;
  const int blocks = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
  hipLaunchKernelGGL(scatter_dense_bits_kernel, dim3(blocks), dim3(256), 0, s, indices, offsets, n_images,
                     total, dims[0], dims[1], dims[2], bits);
  return dn::check_launch("scatter_dense_bits_kernel");
}
