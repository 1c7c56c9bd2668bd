// Common device/host helpers for libcdrl_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace cdrl {

// Channels-last 2-D view: element (row r, channel c) lives at p[r*ld + coff + c].
// A "row" is one pixel of one frame; frames are ordered f = t*B + b so that the rows of
// one BatchNorm statistics group (one time slice) are contiguous.
struct View {
    float* p;
    int ld;
    int coff;
};

__host__ __device__ inline View make_view(float* p, int ld, int coff = 0) {
    View v;
    v.p = p;
    v.ld = ld;
    v.coff = coff;
    return v;
}

enum Act : int { ACT_NONE = 0, ACT_RELU6 = 1, ACT_SWISH6 = 2, ACT_TANH = 3, ACT_SIGMOID = 4 };

// De-interleaving channel shuffle of the reference (core/architectures.py:109-118, F7):
// concat index i -> output channel (i & 1) * (C/2) + (i >> 1).
// ReLU6 pass-through test 0 < z < 6 as ONE vector compare.  min(z, 6 - z) > 0 is exactly that predicate (6 - z is exact near 6 and
// never rounds to 0 unless z == 6; NaN -> false like the two-sided form).  The two-sided form compiles to two v_cmp into scalar
// register pairs combined by s_and_b64 right in front of the v_cndmask that consumes it; see DESIGN.md ("What round 3 found"):
// with another kernel's waves on the same SIMD the top lanes of that mask were occasionally stale.
__device__ __forceinline__ bool relu6_open(float z) { return fminf(z, 6.0f - z) > 0.0f; }

__host__ __device__ inline int shuffle_dst(int i, int ctot) { return (i & 1) * (ctot >> 1) + (i >> 1); }

void set_error(const char* fmt, ...);
const char* last_error();
// getenv for the library's CDRL_* switches: CDRL_DIAG_* (wrong-result timing diagnostics) only with the master CDRL_DIAG=1
const char* cdrl_getenv(const char* name);
int diag_active();                          // number of active CDRL_DIAG_* switches (0 unless CDRL_DIAG=1)
int env_overrides(char* buf, int cap);      // "NAME=VALUE ..." of every CDRL_* variable in the environment; returns their count

#define CDRL_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            cdrl::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -2;                                                                         \
        }                                                                                      \
    } while (0)

#define CDRL_LAUNCH_CHECK() CDRL_HIP(hipGetLastError())

// Tail events (round 6): a fork to another stream used to be hipEventRecord on the critical stream -- a barrier packet of its own in
// the queue, a bubble in front of the next kernel, ~70 times per update-step.  hipExtLaunchKernelGGL binds a STOP EVENT to the kernel's
// own dispatch packet instead (its completion signal), so the other stream can wait for "the newest kernel of the critical stream"
// without anything being enqueued there.  A stop event on EVERY kernel costs more than the bubbles it removes (+0.16 ms per
// update-step), so the launches that need one are LEARNED: while a thread has a TailEvents installed (Learner::launch), launch number i
// of the body carries an event iff a fork followed launch i the last time this body ran (`need`); a fork that finds no event behind the
// newest kernel (`last` null: first run, a changed sequence, a copy or memset behind the kernel) records one the old way and marks the
// launch for the next run.  Always correct; converges after one run of each body.
struct TailEvents {
    hipStream_t stream = nullptr;
    hipEvent_t ring[32] = {};
    int n = 0;
    unsigned head = 0;
    hipEvent_t last = nullptr;      // stop event of the newest kernel on `stream`, if it carries one
    int idx = 0;                    // launches of this body on `stream` so far
    uint8_t* need = nullptr;        // per launch index of the current body: 1 = a fork followed it
    int need_cap = 0;
};
extern thread_local TailEvents* tl_tail;

template <typename... Args, typename... Act>
inline void launch_tracked(void (*kernel)(Args...), dim3 grid, dim3 block, uint32_t lds, hipStream_t st, Act&&... args) {
    TailEvents* t = tl_tail;
    if (t && st == t->stream) {
        const int i = t->idx++;
        if (i < t->need_cap && t->need[i]) {
            hipEvent_t ev = t->ring[t->head++ % (unsigned)t->n];
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, nullptr, ev, 0, static_cast<Args>(args)...);
            t->last = ev;
            return;
        }
        t->last = nullptr;
    }
    kernel<<<grid, block, lds, st>>>(static_cast<Args>(args)...);
}
inline void tail_invalidate(hipStream_t st) {
    TailEvents* t = tl_tail;
    if (t && st == t->stream) t->last = nullptr;
}

// "Set the dynamic-LDS attribute of this kernel once" guards: once PER DEVICE (a second engine on another device of the same process, or
// the optional second enqueue thread racing the first launch, must not launch without it -- ADVICE r4): one flag per device and guard
// site, set after the attribute call returned; the call is idempotent, so two threads doing it at once is harmless.
struct LdsAttrOnce {
    volatile unsigned char done[64] = {};
    bool need() const {
        int d = 0;
        (void)hipGetDevice(&d);
        return !done[d & 63];
    }
    void mark() {
        int d = 0;
        (void)hipGetDevice(&d);
        done[d & 63] = 1;
    }
};

#define CDRL_TRY(expr)            \
    do {                          \
        int _r = (expr);          \
        if (_r != 0) return _r;   \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline int pow2ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// Geometry of the "column-mapped" kernels: blockDim = (CX channel lanes, CY row lanes), 256 threads.
struct ColGeom {
    int cx, cy;     // block dims
    int rb;         // rows per block
    int nb;         // blocks per group
};

// blocks per BatchNorm group for the statistics / backward reductions (x G groups), and for the
// single-group filter-gradient reductions (depthwise / stem): enough workgroups to fill 256 CUs.
// (128 since round 5 -- 256 before: the step is indifferent to it alone, 14.33 vs 14.32 ms, 64 costs +0.48 ms; with 128 rows the finalize
//  folded into the consumer's prologue (CDRL_FIN_ON_LOAD) reads half as much per workgroup: 14.15 vs 14.31 ms same box)
constexpr int NB_STATS = 128;
constexpr int NB_FILTER = 1024;

inline ColGeom col_geom(int rows_per_group, int C, int max_blocks_per_group = NB_STATS) {
    ColGeom g;
    g.cx = pow2ceil(C);
    if (g.cx < 32) g.cx = 32;
    if (g.cx > 256) g.cx = 256;
    g.cy = 256 / g.cx;
    int rb = cdiv(rows_per_group, max_blocks_per_group);
    // many small workgroups: the streaming kernels need >= 1024 resident workgroups to cover HBM
    // latency (measured: 64 elements/thread minimum made the late stages 2-3x slower); the finalize
    // kernels cope with the larger partial counts by summing them with 64 lanes per channel
    int minrb = g.cy * 4;
    if (rb < minrb) rb = minrb;
    rb = cdiv(rb, g.cy) * g.cy;
    g.rb = rb;
    g.nb = cdiv(rows_per_group, rb);
    return g;
}

// Geometry of the VECTORISED column-mapped kernels (BatchNorm family, column sums): every thread
// owns `vec` consecutive channels (16-byte accesses when C % 4 == 0, 8-byte when C is even), the
// block is (cx = C/vec channel lanes) x (cy row lanes) -- not a power of two in general (C = 116:
// 29 x 8 = 232 threads).  A function of (rows, C) only, so producers and consumers of the partial
// sums agree on `nb` without talking to each other.
// TF 'SAME' geometry of a 3x3 window (asymmetric padding for stride 2 on even sizes)
__host__ __device__ inline int same_out(int n, int s) { return (n + s - 1) / s; }
__host__ __device__ inline int same_pad_before(int n, int s) {
    const int out = (n + s - 1) / s;
    int tot = (out - 1) * s + 3 - n;
    if (tot < 0) tot = 0;
    return tot / 2;
}

struct VColGeom {
    int vec, cx, cy, nloop;   // nloop: channel-lane passes when C/vec > 256
    int rb, nb;
};

inline VColGeom vcol_geom(int rows_per_group, int C, int max_blocks_per_group = NB_STATS) {
    VColGeom g;
    g.vec = (C % 4 == 0) ? 4 : ((C % 2 == 0) ? 2 : 1);
    const int lanes = C / g.vec;
    g.cx = lanes > 256 ? 256 : lanes;
    g.nloop = cdiv(lanes, g.cx);
    g.cy = 256 / g.cx;
    if (g.cy < 1) g.cy = 1;
    int rb = cdiv(rows_per_group, max_blocks_per_group);
    int minrb = g.cy * 2;
    if (rb < minrb) rb = minrb;
    rb = cdiv(rb, g.cy) * g.cy;
    g.rb = rb;
    g.nb = cdiv(rows_per_group, rb);
    return g;
}

}  // namespace cdrl

// every launch of the library goes through launch_tracked (see TailEvents above)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) ::cdrl::launch_tracked(kernel, dim3(grid), dim3(block), (uint32_t)(lds), stream, __VA_ARGS__)
