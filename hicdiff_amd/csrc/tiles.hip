// Tile producer / stitcher: a chromosome contact matrix <-> the band of
// upper-triangle piece x piece tiles the samplers consume (processdata/PrepareData_linear_sing.py:25-46).
// Pure data movement, HBM-bound: every kernel here reads and writes each float once, lanes along the
// contiguous axis on both sides (the mirrored half of the stitch goes through an LDS transpose).
#include "hd_common.h"

// One workgroup per (tile, 16-row slab).  Elements past the matrix edge are the zero padding the
// reference adds before cutting (F.pad(..., value=0.0), PrepareData_linear_sing.py:34-38).
template <bool VEC4>
__global__ __launch_bounds__(256) void split_pieces_kernel(const float* __restrict__ mat, int n, const int* __restrict__ origins,
                                                           int piece, float* __restrict__ tiles) {
    const int t = blockIdx.x;
    const int i0 = origins[2 * t], j0 = origins[2 * t + 1];
    float* dst = tiles + (size_t)t * piece * piece;
    if (VEC4 && (j0 & 3) == 0) {                               // workgroup-uniform: a tile whose columns start off a 16-byte boundary goes scalar
        const int q = piece >> 2;                              // float4 per tile row
        for (int e = blockIdx.y * 256 + threadIdx.x; e < piece * q; e += gridDim.y * 256) {
            const int r = e / q, c = (e - r * q) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i0 + r < n && j0 + c < n) v = *reinterpret_cast<const float4*>(mat + (size_t)(i0 + r) * n + j0 + c);
            *reinterpret_cast<float4*>(dst + (size_t)r * piece + c) = v;
        }
    } else {
        for (int e = blockIdx.y * 256 + threadIdx.x; e < piece * piece; e += gridDim.y * 256) {
            const int r = e / piece, c = e - r * piece;
            dst[e] = (i0 + r < n && j0 + c < n) ? mat[(size_t)(i0 + r) * n + j0 + c] : 0.f;
        }
    }
}

// Value of matrix element (r, c) from the tile that holds it directly, or NaN-free "absent" flag.
__device__ __forceinline__ bool tile_fetch(const float* __restrict__ tiles, const int* __restrict__ tile_of, int nb, int piece, int step,
                                           int r, int c, float& v) {
    const int kr = r / step, kc = c / step;
    const int rr = r - kr * step, cc = c - kc * step;
    if (rr >= piece || cc >= piece) return false;
    const int t = tile_of[kr * nb + kc];
    if (t < 0) return false;
    v = tiles[((size_t)t * piece + rr) * piece + cc];
    return true;
}

// One workgroup per 64x64 block of the output matrix.  Blocks on or above the diagonal read their tile
// row-wise; blocks below it read the mirror tile with lanes along its rows (= output columns) into LDS
// and write the transpose, so both the loads and the stores stay coalesced.  A directly held element (only a diagonal
// tile can hold one below the diagonal) wins over the mirror, in both branches.
__global__ __launch_bounds__(256) void stitch_pieces_kernel(const float* __restrict__ tiles, const int* __restrict__ tile_of, int nb, int piece,
                                                            int step, float* __restrict__ mat, int n) {
    __shared__ float lds[64][65];
    const int R0 = blockIdx.y * 64, C0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool lower = R0 >= C0 + 64;
    if (!lower) {
        for (int rr = w; rr < 64; rr += 4) {
            const int r = R0 + rr, c = C0 + lane;
            if (r >= n || c >= n) continue;
            float v = 0.f;
            if (!tile_fetch(tiles, tile_of, nb, piece, step, r, c, v)) tile_fetch(tiles, tile_of, nb, piece, step, c, r, v);
            mat[(size_t)r * n + c] = v;
        }
    } else {
        for (int cc = w; cc < 64; cc += 4) {                  // lanes along the output row index = the mirror tile's column index
            const int r = R0 + lane, c = C0 + cc;
            float v = 0.f;
            if (r < n && c < n) { if (!tile_fetch(tiles, tile_of, nb, piece, step, r, c, v)) tile_fetch(tiles, tile_of, nb, piece, step, c, r, v); }
            lds[cc][lane] = v;
        }
        __syncthreads();
        for (int rr = w; rr < 64; rr += 4) {
            const int r = R0 + rr, c = C0 + lane;
            if (r < n && c < n) mat[(size_t)r * n + c] = lds[lane][rr];
        }
    }
}

// The same with 16-byte accesses, for n, piece and step all multiples of 4: the four elements of an aligned float4 then
// share one tile cell, so one table lookup and one float4 load serve them.  Only a diagonal tile's own lower half
// (direct, upper blocks) and mirrored reads in the diagonal block fall back to per-element gathers.
constexpr int STITCH_STRIP = 8;

__device__ __forceinline__ float4 stitch_quad(const float* __restrict__ tiles, const int* __restrict__ tile_of, int nb, int piece, int step,
                                              int r, int c, bool& held) {
    const int kr = r / step, kc = c / step;
    const int rr = r - kr * step, cc = c - kc * step;
    held = false;
    if (rr >= piece || cc >= piece) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int t = tile_of[kr * nb + kc];
    if (t < 0) return make_float4(0.f, 0.f, 0.f, 0.f);
    held = true;
    return *reinterpret_cast<const float4*>(tiles + ((size_t)t * piece + rr) * piece + cc);
}

__global__ __launch_bounds__(256) void stitch_pieces_vec4_kernel(const float* __restrict__ tiles, const int* __restrict__ tile_of, int nb, int piece,
                                                                 int step, float* __restrict__ mat, int n) {
    __shared__ float lds[64][65];
    const int q = threadIdx.x & 15, row = threadIdx.x >> 4;   // 16 float4 per 64-wide row, 16 rows per pass
    const int R0 = blockIdx.y * 64;
    // one workgroup walks STRIP 64x64 blocks of a row of blocks: most of a chromosome-sized matrix lies outside the band,
    // and a 16 KB zero fill per workgroup would be bound by workgroup dispatch, not by HBM
    for (int sb = 0; sb < STITCH_STRIP; ++sb) {
    const int C0 = (blockIdx.x * STITCH_STRIP + sb) * 64;
    if (C0 >= n) break;
    // does any tile (or mirror tile) touch this block at all?  One step-grid cell per thread, both orientations.
    const int kr0 = R0 / step, kr1 = min(nb - 1, (R0 + 63) / step), kc0 = C0 / step, kc1 = min(nb - 1, (C0 + 63) / step);
    const int ncell = (kr1 - kr0 + 1) * (kc1 - kc0 + 1);
    int found = 0;
    for (int e = threadIdx.x; e < ncell; e += 256) {
        const int kr = kr0 + e / (kc1 - kc0 + 1), kc = kc0 + e % (kc1 - kc0 + 1);
        found |= (tile_of[kr * nb + kc] >= 0) | (tile_of[kc * nb + kr] >= 0);
    }
    if (!__syncthreads_or(found)) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = R0 + pass * 16 + row, c = C0 + 4 * q;
            if (r < n && c < n) *reinterpret_cast<float4*>(mat + (size_t)r * n + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        continue;
    }
    const bool lower = R0 >= C0 + 64;
    if (!lower) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = R0 + pass * 16 + row, c = C0 + 4 * q;
            if (r >= n || c >= n) continue;
            bool held;
            float4 v = stitch_quad(tiles, tile_of, nb, piece, step, r, c, held);
            if (!held) {                                       // mirror: a column of some tile, or nothing
                float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) tile_fetch(tiles, tile_of, nb, piece, step, c + k, r, e[k]);
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
            *reinterpret_cast<float4*>(mat + (size_t)r * n + c) = v;
        }
    } else {
        // a float4 along the OUTPUT ROW index r is a float4 along the mirror tile's row: (c, r..r+3)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int cc = pass * 16 + row, r = R0 + 4 * q, c = C0 + cc;
            float e[4] = {0.f, 0.f, 0.f, 0.f};
            if (r < n && c < n) {
                bool held;
                const float4 v = stitch_quad(tiles, tile_of, nb, piece, step, c, r, held);
                e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
                // a directly held element wins (a diagonal tile reaching below the 64-block diagonal: piece not a multiple of 64)
#pragma unroll
                for (int k = 0; k < 4; ++k) tile_fetch(tiles, tile_of, nb, piece, step, r + k, c, e[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) lds[cc][4 * q + k] = e[k];
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rr = pass * 16 + row, r = R0 + rr, c = C0 + 4 * q;
            if (r < n && c < n)
                *reinterpret_cast<float4*>(mat + (size_t)r * n + c) = make_float4(lds[4 * q][rr], lds[4 * q + 1][rr], lds[4 * q + 2][rr], lds[4 * q + 3][rr]);
        }
        __syncthreads();                                       // the LDS image is reused by the next block of the strip
    }
    }
}

int launch_split_pieces(const float* mat, int n, const int* origins, int ntiles, int piece, float* tiles, hipStream_t st) {
    if (ntiles == 0) return 0;
    const int slabs = std::max(1, std::min(16, piece * piece / 1024));
    dim3 grid(ntiles, slabs);
    // float4 rows need 16-byte aligned sources: the matrix pitch a multiple of 4 floats here, the tile's first column in the kernel.
    const bool vec4 = (n % 4 == 0) && (piece % 4 == 0) && (reinterpret_cast<uintptr_t>(mat) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(tiles) % 16 == 0);
    if (vec4) hipLaunchKernelGGL(split_pieces_kernel<true>, grid, dim3(256), 0, st, mat, n, origins, piece, tiles);
    else hipLaunchKernelGGL(split_pieces_kernel<false>, grid, dim3(256), 0, st, mat, n, origins, piece, tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("split_pieces: ") + hipGetErrorString(e)); return -3; }
    return 0;
}

int launch_stitch_pieces(const float* tiles, const int* tile_of, int nb, int piece, int step, float* mat, int n, hipStream_t st) {
    if (n == 0) return 0;
    const int g = (n + 63) / 64;
    const bool vec4 = (n % 4 == 0) && (piece % 4 == 0) && (step % 4 == 0) && (reinterpret_cast<uintptr_t>(mat) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(tiles) % 16 == 0);
    if (vec4) hipLaunchKernelGGL(stitch_pieces_vec4_kernel, dim3((g + STITCH_STRIP - 1) / STITCH_STRIP, g), dim3(256), 0, st, tiles, tile_of, nb, piece, step, mat, n);
    else hipLaunchKernelGGL(stitch_pieces_kernel, dim3(g, g), dim3(256), 0, st, tiles, tile_of, nb, piece, step, mat, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hd_set_error(std::string("stitch_pieces: ") + hipGetErrorString(e)); return -3; }
    return 0;
}
