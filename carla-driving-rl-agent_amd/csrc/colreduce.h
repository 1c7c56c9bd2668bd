// Column-mapped reduction skeleton shared by the BatchNorm / depthwise / stem kernels.
#pragma once
#include "cdrl_kernels.h"

namespace cdrl {

// ------------------------------------------------------------------------------------------
// column-mapped reduction skeleton
// ------------------------------------------------------------------------------------------
template <int NQ, class F>
__global__ void __launch_bounds__(256) colreduce_kernel(F f, int Mg, int C, int rb, double* __restrict__ part) {
    extern __shared__ double sm[];   // [CY][NQ][CX]
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int g = blockIdx.y;
    const int nb = gridDim.x;
    const int r0 = blockIdx.x * rb;
    const int r1 = min(r0 + rb, Mg);
    for (int c0 = 0; c0 < C; c0 += CX) {
        const int c = c0 + tx;
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
        if (c < C) {
            for (int r = r0 + ty; r < r1; r += CY) f(g, (int64_t)g * Mg + r, c, acc);
        }
        if (CY > 1) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) sm[(ty * NQ + q) * CX + tx] = acc[q];
            __syncthreads();
            if (ty == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    double s = acc[q];
                    for (int y = 1; y < CY; ++y) s += sm[(y * NQ + q) * CX + tx];
                    acc[q] = s;
                }
            }
            __syncthreads();
        }
        if (ty == 0 && c < C) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) part[(((int64_t)g * nb + blockIdx.x) * NQ + q) * C + c] = acc[q];
        }
    }
}

template <int NQ, class F>
static int launch_colreduce(F f, int G, int Mg, int C, double* part, hipStream_t st, int max_blocks = NB_STATS) {
    ColGeom g = col_geom(Mg, C, max_blocks);
    dim3 grid(g.nb, G), block(g.cx, g.cy);
    size_t sm = (size_t)g.cy * NQ * g.cx * sizeof(double);
    hipLaunchKernelGGL((colreduce_kernel<NQ, F>), grid, block, sm, st, f, Mg, C, g.rb, part);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// vectorised skeleton: F::operator()(g, row, c0, acc) handles VEC consecutive channels c0..c0+VEC-1
// of one row and accumulates into acc[NQ][VEC].
// ------------------------------------------------------------------------------------------
template <int VEC>
struct VecF {
    float v[VEC];
};

template <int VEC>
__device__ __forceinline__ VecF<VEC> vload(const float* p) {
    VecF<VEC> r;
    if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2 % VEC] = t.z; r.v[3 % VEC] = t.w;
    } else if (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        r.v[0] = t.x; r.v[1 % VEC] = t.y;
    } else {
        r.v[0] = *p;
    }
    return r;
}

template <int VEC>
__device__ __forceinline__ void vstore(float* p, const VecF<VEC>& r) {
    if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % VEC], r.v[2 % VEC], r.v[3 % VEC]);
    else if (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1 % VEC]);
    else *p = r.v[0];
}

// ---- bf16 ACTIVATION STORAGE (configuration 3): the same kernels read / write activation tensors as bf16 (round to nearest
// even on store, exact widening on load); everything a kernel computes with -- registers, LDS tiles, statistics, partials,
// coefficients, weights -- stays float32 / double.  A tensor's element type is a template parameter T of the kernel (float
// instantiations are the float32 path, unchanged); on the host side the pointers keep their `float*` / View spelling and an
// `at` flag (0: float32, 1: bf16) says what they point to.
typedef __bf16 bf16_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t bf_pack(float a, float b) {
    bf16x2_t h;
    h[0] = (bf16_t)a;
    h[1] = (bf16_t)b;
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = (bf16_t)v; }

template <int VEC>
__device__ __forceinline__ VecF<VEC> vload(const bf16_t* p) {
    VecF<VEC> r;
    if (VEC == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        r.v[0] = bf_lo(t.x); r.v[1 % VEC] = bf_hi(t.x); r.v[2 % VEC] = bf_lo(t.y); r.v[3 % VEC] = bf_hi(t.y);
    } else if (VEC == 2) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
        r.v[0] = bf_lo(t); r.v[1 % VEC] = bf_hi(t);
    } else {
        r.v[0] = (float)*p;
    }
    return r;
}

// RAW loads: the words as they are in memory, widened later by vdecode().  A bf16 vload() widens right behind the load, and when
// the load sits in a conditional block (`if (p < P) v[u] = vload(...)`) the widening sits there too: every load of an unrolled
// batch then waits for its own data before the next one is issued (s_waitcnt vmcnt(0) per load; the float32 instantiation has
// nothing to do on the loaded registers and issues the whole batch).  Kernels that keep several loads in flight use
//     raw[u] = vload_raw<VEC>(p);  ...all loads of the batch...;  vdecode<VEC>(raw[u], (const T*)nullptr);
// float32: vload_raw == vload, vdecode is a no-op.  bf16: VEC = 4 -> words in v[0], v[1]; 2 -> v[0]; 1 -> the 16 bits in v[0].
template <int VEC>
__device__ __forceinline__ VecF<VEC> vload_raw(const float* p) { return vload<VEC>(p); }
template <int VEC>
__device__ __forceinline__ VecF<VEC> vload_raw(const bf16_t* p) {
    VecF<VEC> r;
    if (VEC == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        r.v[0] = __uint_as_float(t.x);
        r.v[1 % VEC] = __uint_as_float(t.y);
    } else if (VEC == 2) {
        r.v[0] = __uint_as_float(*reinterpret_cast<const uint32_t*>(p));
    } else {
        r.v[0] = __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(p));
    }
    return r;
}
template <int VEC>
__device__ __forceinline__ void vdecode(VecF<VEC>&, const float*) {}
template <int VEC>
__device__ __forceinline__ void vdecode(VecF<VEC>& r, const bf16_t*) {
    if (VEC == 4) {
        const uint32_t a = __float_as_uint(r.v[0]), b = __float_as_uint(r.v[1 % VEC]);
        r.v[0] = bf_lo(a); r.v[1 % VEC] = bf_hi(a); r.v[2 % VEC] = bf_lo(b); r.v[3 % VEC] = bf_hi(b);
    } else if (VEC == 2) {
        const uint32_t a = __float_as_uint(r.v[0]);
        r.v[0] = bf_lo(a); r.v[1 % VEC] = bf_hi(a);
    } else {
        r.v[0] = __uint_as_float(__float_as_uint(r.v[0]) << 16);
    }
}

template <int VEC>
__device__ __forceinline__ void vstore(bf16_t* p, const VecF<VEC>& r) {
    if (VEC == 4) *reinterpret_cast<uint2*>(p) = make_uint2(bf_pack(r.v[0], r.v[1 % VEC]), bf_pack(r.v[2 % VEC], r.v[3 % VEC]));
    else if (VEC == 2) *reinterpret_cast<uint32_t*>(p) = bf_pack(r.v[0], r.v[1 % VEC]);
    else *p = (bf16_t)r.v[0];
}

// typed base pointer of a view (the View struct itself is type-erased: p points to T elements, ld / coff count elements)
template <class T>
__device__ __forceinline__ T* vptr(const View& v) { return reinterpret_cast<T*>(v.p); }

// loads VEC channels of a view; `shuffle_ctot` != 0 -> element-wise through the de-interleave map
template <int VEC, class T = float>
__device__ __forceinline__ VecF<VEC> vload_view(const View& v, int64_t row, int c0, int shuffle_ctot, bool aligned) {
    VecF<VEC> r;
    const T* vp = vptr<T>(v);
    if (!shuffle_ctot && aligned) return vload<VEC>(vp + row * v.ld + v.coff + c0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        int cc = v.coff + c0 + i;
        if (shuffle_ctot) cc = shuffle_dst(cc, shuffle_ctot);
        r.v[i] = ldf(vp + row * v.ld + cc);
    }
    return r;
}

template <int VEC, class T = float>
__device__ __forceinline__ void vstore_view(const View& v, int64_t row, int c0, int shuffle_ctot, bool aligned,
                                            const VecF<VEC>& r) {
    T* vp = vptr<T>(v);
    if (!shuffle_ctot && aligned) {
        vstore<VEC>(vp + row * v.ld + v.coff + c0, r);
        return;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        int cc = v.coff + c0 + i;
        if (shuffle_ctot) cc = shuffle_dst(cc, shuffle_ctot);
        stf(vp + row * v.ld + cc, r.v[i]);
    }
}

// gradient w.r.t. the pre-pool activation, gathered from the pooled gradient through the saved argmax
template <int VEC, class T = float>
__device__ __forceinline__ VecF<VEC> pool_gather(const PoolSrc& ps, int64_t row, int c0, int C) {
    const int ix = (int)(row % ps.W);
    const int64_t q = row / ps.W;
    const int iy = (int)(q % ps.H);
    const int64_t n = q / ps.H;
    VecF<VEC> acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.v[i] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ny = iy + ps.pt - ky;
        if (ny < 0 || (ny & 1)) continue;
        const int oy = ny >> 1;
        if (oy >= ps.Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int nx = ix + ps.pl - kx;
            if (nx < 0 || (nx & 1)) continue;
            const int ox = nx >> 1;
            if (ox >= ps.Wo) continue;
            const int64_t o = ((n * ps.Ho + oy) * ps.Wo + ox) * C + c0;
            const VecF<VEC> d = vload<VEC>(reinterpret_cast<const T*>(ps.dp) + o);
            if (VEC == 4) {       // the 4 argmax bytes of this lane in one 32-bit load
                const uint32_t am = *reinterpret_cast<const uint32_t*>(ps.argmax + o);
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (((am >> (8 * i)) & 0x7fu) == (uint32_t)(ky * 3 + kx)) acc.v[i] += d.v[i];      // (bit 7: ReLU6 flag of maxpool_bn_fwd)
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if ((ps.argmax[o + i] & 0x7f) == (uint8_t)(ky * 3 + kx)) acc.v[i] += d.v[i];
            }
        }
    }
    return acc;
}

__host__ inline bool view_aligned(const View& v, int vec) {
    return (v.ld % vec == 0) && (v.coff % vec == 0) && ((reinterpret_cast<uintptr_t>(v.p) % (4 * vec)) == 0);
}

template <int NQ, int VEC, class F>
__global__ void __launch_bounds__(256) vcolreduce_kernel(F f, int Mg, int C, int rb, int nloop,
                                                         double* __restrict__ part) {
    extern __shared__ double sm[];   // [CY][VEC][CX]
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int g = blockIdx.y;
    const int nb = gridDim.x;
    const int r0 = blockIdx.x * rb;
    const int r1 = min(r0 + rb, Mg);
    for (int l = 0; l < nloop; ++l) {
        const int c0 = (l * CX + tx) * VEC;
        double acc[NQ][VEC];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[q][i] = 0.0;
        if (c0 < C) {
            for (int r = r0 + ty; r < r1; r += CY) f(g, (int64_t)g * Mg + r, c0, acc);
        }
        if (CY > 1) {
            // one quantity at a time through a [CY][VEC][CX] LDS slab (keeps LDS at a few KB for NQ = 10)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) sm[(ty * VEC + i) * CX + tx] = acc[q][i];
                __syncthreads();
                if (ty == 0) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        double s = acc[q][i];
                        for (int y = 1; y < CY; ++y) s += sm[(y * VEC + i) * CX + tx];
                        acc[q][i] = s;
                    }
                }
                __syncthreads();
            }
        }
        if (ty == 0 && c0 < C) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    part[(((int64_t)g * nb + blockIdx.x) * NQ + q) * C + c0 + i] = acc[q][i];
        }
    }
}

// F<VEC> must be a class template; launches the variant matching vcol_geom(Mg, C).vec
template <int NQ, template <int> class F, class... Args>
static int launch_vcolreduce(int G, int Mg, int C, double* part, hipStream_t st, int max_blocks, Args... args) {
    VColGeom g = vcol_geom(Mg, C, max_blocks);
    dim3 grid(g.nb, G), block(g.cx, g.cy);
    const size_t sm = (size_t)g.cy * g.vec * g.cx * sizeof(double);
    if (g.vec == 4) {
        F<4> f{args...};
        hipLaunchKernelGGL((vcolreduce_kernel<NQ, 4, F<4>>), grid, block, sm, st, f, Mg, C, g.rb, g.nloop, part);
    } else if (g.vec == 2) {
        F<2> f{args...};
        hipLaunchKernelGGL((vcolreduce_kernel<NQ, 2, F<2>>), grid, block, sm, st, f, Mg, C, g.rb, g.nloop, part);
    } else {
        F<1> f{args...};
        hipLaunchKernelGGL((vcolreduce_kernel<NQ, 1, F<1>>), grid, block, sm, st, f, Mg, C, g.rb, g.nloop, part);
    }
    CDRL_LAUNCH_CHECK();
    return 0;
}

// the same for functors F<VEC, T> over activation tensors of element type T (at: 0 float32, 1 bf16)
template <int NQ, template <int, class> class F, class T, class... Args>
static int launch_vcolreduce_t(int G, int Mg, int C, double* part, hipStream_t st, int max_blocks, Args... args) {
    VColGeom g = vcol_geom(Mg, C, max_blocks);
    dim3 grid(g.nb, G), block(g.cx, g.cy);
    const size_t sm = (size_t)g.cy * g.vec * g.cx * sizeof(double);
    if (g.vec == 4) {
        F<4, T> f{args...};
        hipLaunchKernelGGL((vcolreduce_kernel<NQ, 4, F<4, T>>), grid, block, sm, st, f, Mg, C, g.rb, g.nloop, part);
    } else if (g.vec == 2) {
        F<2, T> f{args...};
        hipLaunchKernelGGL((vcolreduce_kernel<NQ, 2, F<2, T>>), grid, block, sm, st, f, Mg, C, g.rb, g.nloop, part);
    } else {
        F<1, T> f{args...};
        hipLaunchKernelGGL((vcolreduce_kernel<NQ, 1, F<1, T>>), grid, block, sm, st, f, Mg, C, g.rb, g.nloop, part);
    }
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
