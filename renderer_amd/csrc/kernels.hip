// HIP kernels of the frame hot path, written for gfx950 (CDNA4, wave64).
//
//   K1  k_transform   coalesced SoA vertex transform            (render_mesh.wgsl:42-82 vs_main)
//   K2a k_setup       triangle setup: cull, snap, bbox, tile count, wave-compacted record stream
//   K2c k_clip        Sutherland-Hodgman clipper for the (rare) triangles K2a queues
//       k_scan_tiles  exclusive scan of the per-tile counts
//   K2b k_bin_fill    scatter record slots into per-tile bins
//   K3  k_raster      one workgroup per screen tile; depth|id staged in LDS, fixed-point edges
//   K4  k_shade       deferred material/light shading + tone map + sRGB
//                     (render_mesh.wgsl:86-111, :134-181, :256-281; show_texture.wgsl:40-61)
//
// The arithmetic that decides coverage and depth is integer or single IEEE f32 operations in a
// fixed order (compiled with -ffp-contract=off; f32 divide and sqrt are correctly rounded by
// hipcc's default), as DESIGN.md "Raster rules" states it, so results are reproducible bit for bit.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "kernels.h"


namespace rnd {


// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
constexpr uint32_t kLdsItems = 1023;       // draw-item tables up to this size are searched in LDS

// Internal consistency checks: an index that would leave its buffer sets a bit of
// FrameCounters::overflow (reported by rnd_finish as RND_ERR_DEVICE) and the access is skipped,
// instead of faulting the GPU.
enum : uint32_t { kGuardSetup = 0x100u, kGuardCount = 0x200u, kGuardBin = 0x400u, kGuardRaster = 0x800u, kGuardShade = 0x1000u };
__device__ __forceinline__ bool guard(bool ok, FrameCounters* counters, uint32_t bit) {
    if (!ok) atomicOr(&counters->overflow, bit);
    return ok;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Whether the frame is withheld: a pass overflowed an internal stream (records, clip queue, bin entries), or the frame was
// launched without the record chain (k_clip .. k_raster) and a pass turned out to need it.  Wave-uniform scalar loads.
__device__ __forceinline__ bool frame_overflowed(const OverflowGate& g) {
    typedef const __attribute__((address_space(4))) FrameCounters* cptr_counters;
    uint32_t ovf = 0, need = 0;
    for (uint32_t s = 0; s < g.n_counter_sets; ++s) {
        if (s < 64u && !((g.ran_mask >> s) & 1ull)) continue;
        ovf |= ((cptr_counters)g.all_counters + s)->overflow;
        need |= ((cptr_counters)g.all_counters + s)->chain_needed;
    }
    return (ovf & 7u) != 0u || (g.no_chain != 0u && need != 0u);
}

// The frame's constants live in device memory (uploaded with the per-frame tables), not in the kernel
// arguments (one upload per frame, small launch packets; the shading kernel alone takes the frame's shape by value:
// FrameBuffers::dims).  They are read through the constant
// address space — scalar loads the compiler may repeat or re-order like kernel arguments — field by
// field where used (copying the struct up front kept all 70 dwords live: SGPR spills into VGPR lanes).
typedef const __attribute__((address_space(4))) FrameConst& FrameConstRef;
typedef const __attribute__((address_space(4))) float* cptr_f32;
__device__ __forceinline__ FrameConstRef frame_const_ref(const FrameConst* p) {
    return *(const __attribute__((address_space(4))) FrameConst*)p;
}

// largest k in [0, n) with bases[k] <= id (bases[0] == 0, ascending)
__device__ __forceinline__ uint32_t find_item(const uint32_t* __restrict__ bases, uint32_t n, uint32_t id) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (bases[mid] <= id) lo = mid; else hi = mid;
    }
    return lo;
}

// The same for one wave-uniform id and n <= 64, by all lanes at once: lane k looks at [bases[k], bases[k+1]) (bases has
// n + 1 entries), the one lane that holds the id answers.  n when no lane does (id >= bases[n]).
__device__ __forceinline__ uint32_t find_item_wave(const uint32_t* __restrict__ bases, uint32_t n, uint32_t id_uniform, uint32_t lane) {
    bool in = false;
    if (lane < n) in = bases[lane] <= id_uniform && id_uniform < bases[lane + 1];
    const unsigned long long m = __ballot(in);
    return m ? (uint32_t)__ffsll((long long)m) - 1u : n;
}

__device__ __forceinline__ bool owns_tile(uint32_t tx, uint32_t ty, uint32_t rank, uint32_t world) {
    if (world <= 1) return true;
    (void)tx;
    return stripe_owner(ty / kOwnerGroup, world) == rank;   // full-width stripes of kOwnerGroup tile rows
}

// 24-bit signed multiply (+ 32-bit add): full-rate instructions that take the low 24 bits of their operands as they
// are.  Written as assembly because the library routine behind __mul24 sign-extends each operand first (a v_bfe_i32
// apiece that the multiply does not need); plain VALU instructions, no hazards of their own.
__device__ __forceinline__ int32_t mul24(int32_t a, int32_t b) {
    int32_t r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int32_t mad24(int32_t a, int32_t b, int32_t c) {
    int32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}

// Pointers read out of a DrawItem are generic to the compiler (flat_load, which also ties up the
// LDS counter); these casts make the loads through them plain global loads.
typedef const __attribute__((address_space(1))) uint32_t* gptr_u32;
typedef const __attribute__((address_space(1))) float* gptr_f32;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gptr_f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x2* gptr_f32x2;

// A load at a wave-uniform base + a 32-bit byte offset per lane: one scalar base and one vector offset in the
// instruction, where `base[index]` costs a 64-bit shift and a 64-bit add per lane (4-cycle instructions each).  The
// caller guarantees that the offset fits 32 bits.
template <class T>
__device__ __forceinline__ T load_at(const void* base_uniform, uint32_t byte_offset) {
    typedef const __attribute__((address_space(1))) char* gptr_char;
    typedef const __attribute__((address_space(1))) T* gptr_t;
    return *(gptr_t)((gptr_char)base_uniform + byte_offset);
}

// Waterfall helper: picks the value of `v` held by the first lane of `pending` as a scalar.
// Two things make the obvious `cur = readfirstlane(v); if (cur == v) {...}` loop unsafe: the
// optimizer may hoist the (loop-invariant) readfirstlane out of the loop, and inside the
// `cur == v` branch it may substitute the per-lane `v` for `cur`, turning scalar loads into
// per-lane ones.  Here the lane is chosen from the loop-variant `pending` mask, and the value
// used for addressing passes through an empty asm, so no equality with `v` is known for it.
__device__ __forceinline__ uint32_t uniform_pick(uint32_t v, unsigned long long pending, uint32_t& opaque) {
    uint32_t cur = __builtin_amdgcn_readlane(v, __ffsll((long long)pending) - 1);
    opaque = cur;
    asm volatile("" : "+s"(opaque));
    return cur;
}

// Copies one wave-uniform draw item into registers through the constant address space, which the
// compiler always serves with scalar loads (the table is never written while a kernel reads it).
__device__ __forceinline__ DrawItem load_item_uniform(const DrawItem* items, uint32_t index) {
    // `index` must be wave-uniform (see uniform_pick).
    // member-wise, typed copies: punning the struct through integer vectors would let type-based
    // alias analysis treat the float and pointer fields as never written
    typedef const __attribute__((address_space(4))) DrawItem* cptr_item;
    cptr_item p = (cptr_item)items + index;
    DrawItem d;
    d.px = p->px; d.py = p->py; d.pz = p->pz; d.verts = p->verts; d.indices = p->indices;
    d.n_verts = p->n_verts; d.n_tris = p->n_tris; d.vert_base = p->vert_base; d.tri_base = p->tri_base;
#pragma unroll
    for (int k = 0; k < 16; ++k) { d.world[k] = p->world[k]; d.nrm[k] = p->nrm[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { d.fixed_clip[k] = p->fixed_clip[k]; d.base_color[k] = p->base_color[k]; }
    d.flags = p->flags; d.texture = p->texture;
    return d;
}

// vs_main's clip position (render_mesh.wgsl:42-82).  K1 keeps only what triangle setup reads of it (the snapped
// vertex); the clipper, the alpha test and the shading kernel re-derive it from the object-space position they
// gather anyway: same inputs, same operations, same bits (the shading kernel's default is a cheaper form: corner_clip).
// projection * view of view `v` of the pass (FrameConst: multi-view passes)
__device__ __forceinline__ cptr_f32 view_pv(FrameConstRef fc, uint32_t v) { return v == 0u ? fc.pv : fc.pv_more[v - 1u]; }

__device__ __forceinline__ float4 clip_position(const DrawItem& d, cptr_f32 p, float x, float y, float z) {
    float4 c;
    if ((d.flags & kItemBillboardMask) == RND_BILLBOARD_FIXED_SIZE) {
        c = make_float4(d.fixed_clip[0] + x * 0.4f, d.fixed_clip[1] + y * 0.4f, d.fixed_clip[2], d.fixed_clip[3]);
    } else {
        const float* m = d.world;
        float wx = ((m[0] * x + m[4] * y) + m[8] * z) + m[12];
        float wy = ((m[1] * x + m[5] * y) + m[9] * z) + m[13];
        float wz = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
        float ww = ((m[3] * x + m[7] * y) + m[11] * z) + m[15];
        c.x = ((p[0] * wx + p[4] * wy) + p[8] * wz) + p[12] * ww;
        c.y = ((p[1] * wx + p[5] * wy) + p[9] * wz) + p[13] * ww;
        c.z = ((p[2] * wx + p[6] * wy) + p[10] * wz) + p[14] * ww;
        c.w = ((p[3] * wx + p[7] * wy) + p[11] * wz) + p[15] * ww;
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// K1: vertex transform.  One thread per (draw item, vertex); positions are read from three SoA
// streams (4 B/lane, fully coalesced), the snapped vertex is written as one 16-byte store per lane.
// The draw item is made wave-uniform (waterfall over the rare waves that straddle two items) so
// its matrices come in through scalar loads.
// ------------------------------------------------------------------------------------------
// Frame clear: visibility words of the owned tiles to (depth 1.0, no triangle), tile counts and
// frame counters to zero.  One workgroup per owned tile.  The first clear of a frame also brings in
// the frame's tables: `n_vec` 16-byte words from the pinned host block `src` (read over the host
// link by the kernel itself) to the device block `dst` that every later kernel of the frame reads.
// That replaces a hipMemcpyAsync (a blit kernel of its own, ~3 us plus its dependency gaps per
// frame).  The sizes come in as arguments: the frame's constants are part of what is being copied.
__global__ __launch_bounds__(256) void k_clear(FrameBuffers fb, uint32_t n_counts, uint32_t n_vblocks, uint32_t n_planes,
                                               const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n_vec) {
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n_vec; k += gridDim.x * 256u) dst[k] = src[k];
    if (blockIdx.x < fb.n_owned) {
        const uint32_t tile = fb.launch_order[blockIdx.x].tile;
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 clear = {kClearKey, kClearKey};
        for (uint32_t smp = 0; smp < n_planes; ++smp) {          // (MSAA: one plane per sample)
            u64x2* dst = reinterpret_cast<u64x2*>(fb.vis + (size_t)smp * fb.vis_stride + (size_t)tile * kTilePixels);
            for (uint32_t k = threadIdx.x; k < kTilePixels / 2; k += 256u) dst[k] = clear;
        }
    }
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n_counts; k += gridDim.x * 256u) fb.tile_count[k] = 0;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n_vblocks; k += gridDim.x * 256u) fb.vblock_needed[k] = 0;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(FrameCounters) / 4) reinterpret_cast<uint32_t*>(fb.counters)[threadIdx.x] = 0;
    if (blockIdx.x == 0) for (uint32_t k = threadIdx.x; k < kStatStripes * kStatStride; k += 256u) fb.stat_stripes[k] = 0ull;
}

// Work-unit culling: one thread per (draw item, chunk).  The chunk's object-space box is taken
// through the item's vs_main transform; the unit is dropped when all eight corners lie outside one
// plane of the view volume (with a margin far above the rounding differences to K1's per-vertex
// results), or, on a sort-first partition, when the rows the box can cover (padded by two pixels)
// hold no tile this context owns.  Survivors are appended to active_units and flag the vertex
// blocks they read.
__global__ __launch_bounds__(256) void k_cull(FrameBuffers fb) {
    FrameConstRef fc = frame_const_ref(fb.frame_const);
    const uint32_t gu = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = lane_id();
    bool active = false;
    uint32_t it = 0, chunk = 0, vb0 = 0, vb1 = 0;
    uint4 bounds = make_uint4(0u, 0u, 0u, 0u);
    // early-depth frames in two passes: the units of the far items are only candidates here (with bounds); k_cull_far takes
    // them up again when the near items are drawn
    const bool want_bounds = fb.early_z != 0u && fb.setup_pass != 0u && fb.unit_bounds != nullptr && fb.far_units != nullptr;
    bool far_item = false;
    // (unit, view): the views of a multi-view pass are culled one by one against their own volume
    const uint32_t per_view = max(fc.n_units_view, 1u);
    const uint32_t view = gu / per_view, u = gu - view * per_view;
    const uint32_t n_vblocks = (fc.n_verts + kVertexBlock - 1) / kVertexBlock;
    if (gu < fc.n_units && ((fc.view_mask >> view) & 1u)) {
        // (ranges of units are numbered in processing order — near items first in frames with depth complexity —,
        // the draw item of range k follows the n_items + 1 bases)
        it = fb.unit_bases[fc.n_items + 1u + find_item(fb.unit_bases, fc.n_items, u)];
        const DrawItem& d = fb.items[it];
        chunk = u - d.unit_base;
        far_item = want_bounds && d.unit_base >= fb.split_units;
        const uint32_t flags = d.flags;
        if (!(flags & kItemSkip) && guard(chunk < d.n_chunks, fb.counters, kGuardSetup)) {
            const ChunkInfo ci = d.chunks[chunk];
            vb0 = (d.vert_base + ci.vmin) / kVertexBlock;
            vb1 = (d.vert_base + ci.vmax) / kVertexBlock;
            if ((flags & kItemBillboardMask) == RND_BILLBOARD_FIXED_SIZE) {
                active = true;      // positions do not go through projection * view
            } else {
                bool out_l = true, out_r = true, out_b = true, out_t = true, out_n = true, out_f = true, w_pos = true;
                float y_lo = 3.0e38f, y_hi = -3.0e38f, x_lo = 3.0e38f, x_hi = -3.0e38f, z_lo = 3.0e38f;
                for (uint32_t c = 0; c < 8; ++c) {
                    const float4 p = clip_position(d, view_pv(fc, view), (c & 1u) ? ci.hi[0] : ci.lo[0], (c & 2u) ? ci.hi[1] : ci.lo[1],
                                                   (c & 4u) ? ci.hi[2] : ci.lo[2]);
                    const float e = 1.0e-3f * (fabsf(p.x) + fabsf(p.y) + fabsf(p.z) + fabsf(p.w)) + 1.0e-30f;
                    out_l = out_l && (p.x + p.w < -e); out_r = out_r && (p.w - p.x < -e);
                    out_b = out_b && (p.y + p.w < -e); out_t = out_t && (p.w - p.y < -e);
                    out_n = out_n && (p.z < -e);       out_f = out_f && (p.w - p.z < -e);
                    const bool finite = fabsf(p.x) + fabsf(p.y) + fabsf(p.z) + fabsf(p.w) < 3.0e38f;
                    w_pos = w_pos && finite && p.w > e;
                    const float yn = p.y / p.w;
                    y_lo = fminf(y_lo, yn); y_hi = fmaxf(y_hi, yn);
                    if (want_bounds) { const float xn = p.x / p.w; x_lo = fminf(x_lo, xn); x_hi = fmaxf(x_hi, xn); z_lo = fminf(z_lo, p.z / p.w); }
                }
                active = !(out_l || out_r || out_b || out_t || out_n || out_f);
                if (active && w_pos && far_item) {
                    // Every vertex of the chunk lies in its box; with all eight corners in front of the eye the box's image is
                    // the convex hull of the corners' images, so the corners bound the chunk's pixels (two pixels of slack for
                    // snapping and rounding, as below) and its depth (z/w takes its extremes at corners; 4e-6 of slack, ten
                    // times the rounding of K1's routine, which evaluates the corners here as it does the vertices there: the layers of C4 lie
                    // 6e-5 apart in depth, a slack of 1e-4 turned none of them away).  k_setup's second early-depth pass drops the whole unit when
                    // every 8x8 block of the rect already holds something nearer everywhere.
                    const float half_w = 0.5f * (float)fc.width, half_h = 0.5f * (float)fc.view_h;
                    const float sx0 = fmaxf((x_lo + 1.0f) * half_w - 2.0f, 0.0f), sx1 = fminf((x_hi + 1.0f) * half_w + 2.0f, (float)fc.width - 1.0f);
                    const float sy0 = fmaxf((1.0f - y_hi) * half_h - 2.0f, 0.0f), sy1 = fminf((1.0f - y_lo) * half_h + 2.0f, (float)fc.view_h - 1.0f);
                    if (sx0 <= sx1 && sy0 <= sy1)
                        bounds = make_uint4(((uint32_t)sx0 >> 3) | (((uint32_t)sy0 >> 3) << 16), ((uint32_t)sx1 >> 3) | (((uint32_t)sy1 >> 3) << 16),
                                            __float_as_uint(z_lo - 4.0e-6f), 1u);
                }
                if (active && w_pos && fc.world > 1) {
                    // rows the box can reach, two pixels of slack for snapping and rounding
                    const float half_h = 0.5f * (float)fc.height;
                    const float sy0 = (1.0f - y_hi) * half_h - 2.0f, sy1 = (1.0f - y_lo) * half_h + 2.0f;
                    if (sy1 < 0.0f || sy0 >= (float)fc.height) active = false;
                    else {
                        const uint32_t ty0 = (uint32_t)fmaxf(sy0, 0.0f) / kTile;
                        const uint32_t ty1 = min((uint32_t)fminf(sy1, (float)fc.height), fc.height - 1u) / kTile;
                        const uint32_t g0 = ty0 / kOwnerGroup, g1 = ty1 / kOwnerGroup;
                        bool owned = g1 - g0 + 1u >= fc.world;
                        for (uint32_t g = g0; !owned && g <= g1; ++g) owned = stripe_owner(g, fc.world) == fc.rank;
                        active = owned;
                    }
                }
            }
        }
    }
    const unsigned long long fm = __ballot(active && far_item);
    if (fm) {
        const int leader = __ffsll((long long)fm) - 1;
        uint32_t base = 0;
        if (lane == (uint32_t)leader) base = atomicAdd(&fb.counters->n_far, (uint32_t)__popcll(fm));
        base = __shfl(base, leader);
        if (active && far_item) {
            const uint32_t slot = base + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
            fb.far_units[slot] = make_uint2(it, chunk);
            fb.unit_bounds[slot] = bounds;
        }
        active = active && !far_item;
    }
    const unsigned long long am = __ballot(active);
    if (am) {
        const int leader = __ffsll((long long)am) - 1;
        uint32_t base = 0;
        if (lane == (uint32_t)leader) base = atomicAdd(&fb.counters->n_active, (uint32_t)__popcll(am));
        base = __shfl(base, leader);
        if (active) {
            fb.active_units[base + (uint32_t)__popcll(am & ((1ull << lane) - 1ull))] = make_uint2(it, chunk | (view << kUnitViewShift));
            if (vb1 - vb0 >= 64u) fb.counters->all_verts = 1u;
            // (the flag names a draw item that has vertices in the block: k_transform starts from it instead of searching)
            else for (uint32_t b = vb0; b <= vb1 && b < n_vblocks; ++b) fb.vblock_needed[view * n_vblocks + b] = it + 1u;
        }
    }
}

constexpr int32_t kSamplePad = 96;      // MSAA x4: the samples sit up to this many sub-pixel steps from the pixel centre
struct Snapped { int32_t x, y; float z; };
__device__ __forceinline__ bool snap_vertex(float cx, float cy, float cz, float cw, float half_w,
                                            float half_h, Snapped& s, int32_t sub_dx, int32_t sub_dy);

// The per-vertex half of triangle setup: snap (rule 4) and the vertex's verdicts against the view
// volume and the guard band (rules 2, 3).  For finite a, b the rounded sum a + b has the sign of the
// exact sum (and is zero only if that is), so x < -w  <=>  x + w < 0 and x > w  <=>  w - x < 0.
// (kGuardBand is a power of two: kGuardBand * w is exact.)
__device__ __forceinline__ SnapVert snap_and_classify(const float4& c, float half_w, float half_h) {
    Snapped s;
    const bool front = snap_vertex(c.x, c.y, c.z, c.w, half_w, half_h, s, 0, 0);
    const bool finite = isfinite(c.x) && isfinite(c.y) && isfinite(c.z) && isfinite(c.w);
    const float g = kGuardBand * c.w, zf = c.w - c.z;
    uint32_t code = (c.x + c.w < 0.0f ? 1u : 0u) | (c.w - c.x < 0.0f ? 2u : 0u) | (c.y + c.w < 0.0f ? 4u : 0u) |
                    (c.w - c.y < 0.0f ? 8u : 0u) | (c.z < 0.0f ? 16u : 0u) | (zf < 0.0f ? 32u : 0u);
    const bool clip = c.z < 0.0f || zf < 0.0f || c.x + g < 0.0f || g - c.x < 0.0f || c.y + g < 0.0f || g - c.y < 0.0f;
    code |= (clip ? kOutNeedsClip : 0u) | (finite ? 0u : kOutNonFinite) | (front ? 0u : kOutBehind);
    return SnapVert{s.x, s.y, s.z, code};
}

// kBlocks: vertex blocks per workgroup.  1 where most blocks have work.  The second launch of an early-depth frame (the
// far units k_cull_far kept: at C4 three of 44 k) finds nearly every flag clear, and ~100 k workgroups that read a flag
// and exit were 30 us; there a workgroup takes 8 consecutive blocks, their flags in one round trip.  (On a sort-first
// partition, where an eighth of the blocks has work, 8 blocks per workgroup was slower: 15.0 -> 18.7 us.)
template <uint32_t kBlocks>
__global__ __launch_bounds__(256) void k_transform(const DrawItem* __restrict__ items,
                                                   const uint32_t* __restrict__ vert_bases, const FrameConst* __restrict__ fcp,
                                                   float4* __restrict__ clip, SnapVert* __restrict__ snap,
                                                   const uint32_t* __restrict__ vblock_needed,
                                                   const FrameCounters* __restrict__ counters) {
    // (vblock_needed: this launch's row of flags — the far units' in the second launch of an early-depth frame)
    FrameConstRef fc = frame_const_ref(fcp);
    static_assert(kVertexBlock == 256, "one workgroup per vertex block");
    const uint32_t view = blockIdx.y;                                    // multi-view passes: one grid row per view
    if (!((fc.view_mask >> view) & 1u)) return;
    const uint32_t n_vblocks = (fc.n_verts + kVertexBlock - 1u) / kVertexBlock;       // (k_cull's row length of the flags)
    // the blocks' flags (written by k_cull: 1 + a draw item that has vertices here) and the "every block" switch together
    uint32_t hints[kBlocks];
#pragma unroll
    for (uint32_t k = 0; k < kBlocks; ++k) {
        const uint32_t block = blockIdx.x * kBlocks + k;
        hints[k] = block < n_vblocks ? __builtin_amdgcn_readfirstlane(vblock_needed[view * n_vblocks + block]) : 0u;
    }
    const uint32_t all_verts = counters->all_verts;
    const size_t out_base = (size_t)view * fc.n_verts;
    const int32_t view_dy = (int32_t)(view * fc.view_h) * kSub;          // the view's rows in the stacked target
#pragma unroll
    for (uint32_t k = 0; k < kBlocks; ++k) {
        const uint32_t hint_s = hints[k];
        const uint32_t block = blockIdx.x * kBlocks + k;
        if (block >= n_vblocks || (!hint_s && !all_verts)) continue;      // no surviving work unit reads this block
        const uint32_t gid = block * 256u + threadIdx.x;
        if (gid >= fc.n_verts) continue;
        auto transform = [&](const DrawItem& d) {
            uint32_t v = gid - d.vert_base;
            float x = ((gptr_f32)d.px)[v], y = ((gptr_f32)d.py)[v], z = ((gptr_f32)d.pz)[v];
            float4 c = clip_position(d, view_pv(fc, view), x, y, z);
            // (kept for the shading kernel of frames with few vertices per pixel; null otherwise and in depth-only passes:
            // see corner_clip)
            if (clip) clip[gid] = c;
            const SnapVert sv = snap_and_classify(c, 0.5f * (float)fc.width, 0.5f * (float)fc.view_h);
            *reinterpret_cast<uint4*>(snap + out_base + gid) = make_uint4((uint32_t)sv.x, (uint32_t)(sv.y + view_dy), __float_as_uint(sv.z), sv.code);
        };
        // The usual block lies inside one draw item, the one the flag names: no search (the binary search over the items'
        // vertex bases is a chain of dependent loads, which is what this kernel's 64-vertex waves cannot hide).
        unsigned long long pending = __ballot(true);
        if (hint_s && hint_s <= fc.n_items) {
            const DrawItem d = load_item_uniform(items, hint_s - 1u);
            const bool mine = gid - d.vert_base < d.n_verts;         // (unsigned: also false below the item's first vertex)
            if (mine) transform(d);
            pending &= ~__ballot(mine);
        }
        if (!pending) continue;
        const bool left = (pending >> lane_id()) & 1ull;
        uint32_t it = left ? find_item(vert_bases, fc.n_items, gid) : 0u;
        while (pending) {
            uint32_t cur_s, cur = uniform_pick(it, pending, cur_s);
            const bool mine = left && it == cur;
            pending &= ~__ballot(mine);
            if (mine) {
                const DrawItem d = load_item_uniform(items, cur_s);
                transform(d);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2: triangle setup
// ------------------------------------------------------------------------------------------
// (sub_dx, sub_dy): whole sub-pixel steps added after the snap.  With MSAA, sample k of a pixel at
// offset (ox, oy) from the pixel's corner is the pixel centre of the geometry moved by
// (1/2 - ox, 1/2 - oy); the offsets of the 4x pattern are multiples of 1/16 px, exact on the grid.
__device__ __forceinline__ bool snap_vertex(float cx, float cy, float cz, float cw, float half_w,
                                            float half_h, Snapped& s, int32_t sub_dx, int32_t sub_dy) {
    // straight-line; a vertex with w <= 0 yields defined garbage and false (callers drop the triangle).
    // The clamp only matters for that garbage: inside the guard band |s * 256| stays below 2^26.
    float rw = 1.0f / cw;
    float nx = cx * rw, ny = cy * rw, nz = cz * rw;
    float sx = (nx + 1.0f) * half_w;
    float sy = (1.0f - ny) * half_h;
    s.x = (int32_t)fminf(fmaxf(rintf(sx * (float)kSub), -1.0e9f), 1.0e9f) + sub_dx;
    s.y = (int32_t)fminf(fmaxf(rintf(sy * (float)kSub), -1.0e9f), 1.0e9f) + sub_dy;
    s.z = nz;
    return cw > 0.0f;
}

struct TileRect {
    uint32_t tx0, ty0, tx1, ty1;     // tiles touched
    int32_t i0, i1, j0, j1;          // pixels whose centres lie in the bbox (clamped to the frame)
};

// Cull + bbox + record of one snapped triangle.  Returns false when nothing can be covered.
// j_lo .. j_hi: the pixel rows of the triangle's view (its viewport and scissor; the whole frame in single-view passes)
__device__ __forceinline__ bool make_record(const Snapped& a, const Snapped& b, const Snapped& c,
                                            uint32_t tri_id, FrameConstRef fc, int32_t j_lo, int32_t j_hi, SetupRecord& r,
                                            TileRect& tr) {
    long long A2 = (long long)(b.x - a.x) * (c.y - a.y) - (long long)(c.x - a.x) * (b.y - a.y);
    long long area2 = -A2;
    // straight-line: everything is computed, the verdict is one predicate at the end (early returns
    // here became nested branches that each re-initialized the caller's outputs)
    int32_t xmin = min(a.x, min(b.x, c.x)), xmax = max(a.x, max(b.x, c.x));
    int32_t ymin = min(a.y, min(b.y, c.y)), ymax = max(a.y, max(b.y, c.y));
    // pixel centres i*256+128 inside the closed bbox; with MSAA, pixels with any sample position inside it
    // (the samples sit up to kSamplePad sub-pixel steps from the centre)
    const int32_t pad = fc.n_samples > 1u ? kSamplePad : 0;
    int32_t i0 = max((xmin - pad - kSub / 2 + kSub - 1) >> kSubpixelBits, 0);
    int32_t i1 = min((xmax + pad - kSub / 2) >> kSubpixelBits, (int32_t)fc.width - 1);
    int32_t j0 = max((ymin - pad - kSub / 2 + kSub - 1) >> kSubpixelBits, j_lo);
    int32_t j1 = min((ymax + pad - kSub / 2) >> kSubpixelBits, j_hi);
    r.x0 = a.x; r.y0 = a.y; r.x1 = b.x; r.y1 = b.y; r.x2 = c.x; r.y2 = c.y;
    r.z0 = a.z; r.dz1 = b.z - a.z; r.dz2 = c.z - a.z;
    r.inv_area = 1.0f / (float)area2;
    r.tri_id = tri_id;
    r.flags = ((xmax - xmin) < (1 << 14) && (ymax - ymin) < (1 << 14)) ? 1u : 0u;
    tr.tx0 = (uint32_t)i0 / kTile; tr.tx1 = (uint32_t)i1 / kTile;
    tr.ty0 = (uint32_t)j0 / kTile; tr.ty1 = (uint32_t)j1 / kTile;
    tr.i0 = i0; tr.i1 = i1; tr.j0 = j0; tr.j1 = j1;
    return area2 > 0 && i0 <= i1 && j0 <= j1;
}

__device__ __forceinline__ uint32_t pack_tileword(const TileRect& t) {
    return t.tx0 | (t.ty0 << 8) | ((t.tx1 - t.tx0) << 16) | ((t.ty1 - t.ty0) << 24);
}

__device__ __forceinline__ bool any_owned(const TileRect& t, FrameConstRef fc) {
    if (fc.world <= 1) return true;
    // rects spanning a whole ownership period in x or y always contain an owned tile
    if (t.ty1 - t.ty0 >= kOwnerGroup * fc.world) return true;
    for (uint32_t ty = t.ty0; ty <= t.ty1; ++ty)
        if (owns_tile(t.tx0, ty, fc.rank, fc.world)) return true;
    return false;
}

constexpr uint32_t kSmallPixels = 16;

// textureSample with the reference's "filtered" sampler (backend.rs:414-426: Repeat, linear
// mag/min/mip), restated isotropically as in the oracle: rho = longer pixel footprint in texels,
// lambda = log2(rho) clamped to [0, mips-1], bilinear taps at floor(x - 0.5), two levels blended.
// Texel fetchers.  Registered textures are kept as the caller's RGBA8 bytes (4 B per texel, as the reference uploads
// them: backend.rs:365-383) and decoded at the tap through a 512-entry table: [0, 256) the sRGB EOTF (the
// Rgba8UnormSrgb view, evaluated in f64 and rounded once), [256, 512) alpha / 255 — the values the RGBA f32 copy
// of round 1 held, at a quarter of its footprint and tap bytes.  The font atlas of the 2D pass stays RGBA f32.
struct FetchF32 {
    gptr_f32x4 base;
    __device__ __forceinline__ f32x4 operator()(size_t k) const { return base[k]; }
};
template <class Lut>
struct FetchSrgb8 {
    gptr_u32 base;
    Lut lut;
    __device__ __forceinline__ f32x4 operator()(size_t k) const {
        const uint32_t p = base[k];
        return (f32x4){lut[p & 255u], lut[(p >> 8) & 255u], lut[(p >> 16) & 255u], lut[256u + (p >> 24)]};
    }
};

template <class Fetch>
__device__ __forceinline__ f32x4 texture_bilinear(const TextureDesc& t, uint32_t level, float u, float v, const Fetch& fetch_level0) {
    const uint32_t wl = max(1u, t.width >> level), hl = max(1u, t.height >> level);
    size_t off = 0;
    for (uint32_t l = 0; l < level; ++l) off += (size_t)max(1u, t.width >> l) * max(1u, t.height >> l);
    const float x = u * (float)wl - 0.5f, y = v * (float)hl - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    // Repeat: the texel index modulo the level's size.  For a power-of-two size (the rule; uniform over the wave) the
    // float form is exact for every finite f — f / n and n * floor(.) are scalings, the difference is below n — and
    // replaces a 64-bit integer remainder (software, a hundred instructions per coordinate).
    auto wrap = [](float f, uint32_t n) {
        if ((n & (n - 1u)) == 0u) {
            const float fn = (float)n;
            return (uint32_t)(f - fn * floorf(f / fn));
        }
        long long k = (long long)f % (long long)n;
        return (uint32_t)(k < 0 ? k + (long long)n : k);
    };
    const uint32_t x0 = wrap(xf, wl), x1 = wrap(xf + 1.0f, wl), y0 = wrap(yf, hl), y1 = wrap(yf + 1.0f, hl);
    const f32x4 t00 = fetch_level0(off + (size_t)y0 * wl + x0), t10 = fetch_level0(off + (size_t)y0 * wl + x1);
    const f32x4 t01 = fetch_level0(off + (size_t)y1 * wl + x0), t11 = fetch_level0(off + (size_t)y1 * wl + x1);
    const f32x4 top = t00 * (1.0f - fx) + t10 * fx, bot = t01 * (1.0f - fx) + t11 * fx;
    return top * (1.0f - fy) + bot * fy;
}
// `lut`: the 512-entry decode table (global memory, or the shading kernel's copy in LDS)
template <class Lut>
__device__ __forceinline__ f32x4 texture_sample(const TextureDesc& t, Lut lut, float u, float v, float dudx, float dvdx,
                                                float dudy, float dvdy) {
    const FetchSrgb8<Lut> fetch{(gptr_u32)t.texels, lut};
    const float ax = dudx * (float)t.width, ay = dvdx * (float)t.height, bx = dudy * (float)t.width, by = dvdy * (float)t.height;
    const float rho2 = fmaxf(ax * ax + ay * ay, bx * bx + by * by);
    float lambda = 0.5f * log2f(rho2);
    lambda = (lambda != lambda) ? 0.0f : fminf(fmaxf(lambda, 0.0f), (float)(t.mips - 1u));
    const float lf = floorf(lambda), frac = lambda - lf;
    const uint32_t l0 = (uint32_t)lf, l1 = min(l0 + 1u, t.mips - 1u);
    f32x4 a = texture_bilinear(t, l0, u, v, fetch);
    if (frac > 0.0f) {
        const f32x4 b = texture_bilinear(t, l1, u, v, fetch);
        a = a * (1.0f - frac) + b * frac;
    }
    return a;
}

// The alpha test of a textured fragment (render_mesh.wgsl:89-92, :142-146: base = material.base_color *
// textureSample(...); base.a < 0.5 discards), for draw items whose texture has alpha below 1.  Runs
// in the raster stage, before the fragment may enter the (depth, id) resolve: the triangle's clip
// positions (made again from the vertices) and uv are fetched from the id, barycentrics and the uv footprint as at shade time.
__device__ __forceinline__ bool alpha_pass(const FrameBuffers& fb, FrameConstRef fc, uint32_t id, uint32_t i, uint32_t j) {
    const uint32_t it = find_item(fb.tri_bases, fc.n_items, id);
    const DrawItem& d = fb.items[it];
    const uint32_t tex = d.texture;
    if (tex == RND_NONE) return !(d.base_color[3] < 0.5f);
    uint32_t t = id - d.tri_base;
    if (t >= d.n_tris || d.n_verts == 0u) return true;
    gptr_u32 ix = (gptr_u32)d.indices + 3ull * t;
    const uint32_t last_v = d.n_verts - 1u;
    const uint32_t i0 = min(ix[0], last_v), i1 = min(ix[1], last_v), i2 = min(ix[2], last_v);
    gptr_f32x4 vp = (gptr_f32x4)d.verts;
    const f32x4 a0 = vp[2ull * i0], b0v = vp[2ull * i1], c0v = vp[2ull * i2];
    const f32x4 a1 = vp[2ull * i0 + 1], b1v = vp[2ull * i1 + 1], c1v = vp[2ull * i2 + 1];
    // (the corners' clip positions by K1's routine from the positions the vertices bring along: the bits K1 made)
    const float4 c0 = clip_position(d, fc.pv, a0.x, a0.y, a0.z), c1 = clip_position(d, fc.pv, b0v.x, b0v.y, b0v.z),
                 c2 = clip_position(d, fc.pv, c0v.x, c0v.y, c0v.z);
    const float u0 = a1.z, v0 = a1.w, u1 = b1v.z, v1 = b1v.w, u2 = c1v.z, v2 = c1v.w;
    const float xn = fmaf((float)i + 0.5f, fc.ndc_step_xf, -1.0f), yn = fmaf(-((float)j + 0.5f), fc.ndc_step_yf, 1.0f);
    const float px0 = fmaf(-xn, c0.w, c0.x), py0 = fmaf(-yn, c0.w, c0.y);
    const float px1 = fmaf(-xn, c1.w, c1.x), py1 = fmaf(-yn, c1.w, c1.y);
    const float px2 = fmaf(-xn, c2.w, c2.x), py2 = fmaf(-yn, c2.w, c2.y);
    const float f0 = fmaf(px1, py2, -(px2 * py1)), f1 = fmaf(px2, py0, -(px0 * py2)), f2 = fmaf(px0, py1, -(px1 * py0));
    const float S = (f0 + f1) + f2;
    const float is = 1.0f / S;
    float b0 = f0 * is, b1 = f1 * is, b2 = f2 * is;
    if (!(fabsf(b0) + fabsf(b1) + fabsf(b2) < 3.0e38f)) { b0 = 1.0f; b1 = 0.0f; b2 = 0.0f; }
    const float ax0 = c2.w * py1 - c1.w * py2, ax1 = c0.w * py2 - c2.w * py0, ax2 = c1.w * py0 - c0.w * py1;
    const float ay0 = px2 * c1.w - px1 * c2.w, ay1 = px0 * c2.w - px2 * c0.w, ay2 = px1 * c0.w - px0 * c1.w;
    const float Sx = (ax0 + ax1) + ax2, Sy = (ay0 + ay1) + ay2;
    const float is2 = is * is, kx = is2 * fc.ndc_step_xf, ky = is2 * (-fc.ndc_step_yf);
    float dx0 = (ax0 * S - f0 * Sx) * kx, dx1 = (ax1 * S - f1 * Sx) * kx, dx2 = (ax2 * S - f2 * Sx) * kx;
    float dy0 = (ay0 * S - f0 * Sy) * ky, dy1 = (ay1 * S - f1 * Sy) * ky, dy2 = (ay2 * S - f2 * Sy) * ky;
    if (!(fabsf(dx0) + fabsf(dx1) + fabsf(dx2) + fabsf(dy0) + fabsf(dy1) + fabsf(dy2) < 3.0e38f)) dx0 = dx1 = dx2 = dy0 = dy1 = dy2 = 0.0f;
    const float uu = (b0 * u0 + b1 * u1) + b2 * u2, vv = (b0 * v0 + b1 * v1) + b2 * v2;
    const float dudx = (dx0 * u0 + dx1 * u1) + dx2 * u2, dvdx = (dx0 * v0 + dx1 * v1) + dx2 * v2;
    const float dudy = (dy0 * u0 + dy1 * u1) + dy2 * u2, dvdy = (dy0 * v0 + dy1 * v1) + dy2 * v2;
    const f32x4 texel = texture_sample(fb.textures[tex], (gptr_f32)fb.srgb_lut, uu, vv, dudx, dvdx, dudy, dvdy);
    return !(d.base_color[3] * texel.w < 0.5f);
}

// fragment sinks: where a covered sample's (depth, triangle id) key is min-resolved
// (`smp`: the MSAA sample whose plane takes the fragment; 0 without MSAA)
struct LdsSink {
    unsigned long long* s_vis;          // kTilePixels words per sample plane
    int32_t tile_x0, tile_y0;
    __device__ __forceinline__ void operator()(int32_t i, int32_t j, float z, uint32_t id, uint32_t smp = 0) const {
        z = fminf(fmaxf(z, 0.0f), 1.0f);
        if (z < 1.0f) {
            unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | id;
            atomicMin(&s_vis[smp * kTilePixels + (uint32_t)((j - tile_y0) * (int32_t)kTile + (i - tile_x0))], key);
        }
    }
};

// MSAA x4, one geometry pass: the standard pattern's sample positions relative to the pixel centre, in sub-pixel
// steps: (-32,-96) (96,-32) (-96,32) (32,96), i.e. samples 3 and 2 mirror samples 0 and 1.  An edge function
// F = ex*(py - y) - ey*(px - x) moves by ex*dy - ey*dx from the centre to a sample: exact integers, the values the
// per-sample passes of round 1 got by shifting the geometry instead.
template <class W>
struct SampleSteps {
    W a0, a1, a2, b0, b1, b2;           // per edge: step to sample 0 (a) and to sample 1 (b); samples 3 / 2 take -a / -b
};
// (W = int32_t for records below 64 px, whose edge coefficients are below 2^14; int64 otherwise)
template <class W>
__device__ __forceinline__ SampleSteps<W> sample_steps(int32_t ex0, int32_t ey0, int32_t ex1, int32_t ey1, int32_t ex2, int32_t ey2) {
    return SampleSteps<W>{(W)ex0 * -96 - (W)ey0 * -32, (W)ex1 * -96 - (W)ey1 * -32, (W)ex2 * -96 - (W)ey2 * -32,
                          (W)ex0 * -32 - (W)ey0 * 96, (W)ex1 * -32 - (W)ey1 * 96, (W)ex2 * -32 - (W)ey2 * 96};
}
// The fragments of one pixel whose centre has edge values w0..w2: the centre itself, or the four samples.
// `pass()` is the per-pixel alpha test (evaluated once, only if some sample is covered).
template <int kSamples, class W, class Pass, class Emit>
__device__ __forceinline__ void pixel_samples(W w0, W w1, W w2, W b0, W b1, W b2, const SampleSteps<W>& st, Pass&& pass, Emit&& emit) {
    if constexpr (kSamples == 1) {
        if (w0 >= b0 && w1 >= b1 && w2 >= b2 && pass()) emit(w1, w2, 0u);
    } else {
        const W s0[4] = {w0 + st.a0, w0 + st.b0, w0 - st.b0, w0 - st.a0};
        const W s1[4] = {w1 + st.a1, w1 + st.b1, w1 - st.b1, w1 - st.a1};
        const W s2[4] = {w2 + st.a2, w2 + st.b2, w2 - st.b2, w2 - st.a2};
        bool in[4], any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) { in[k] = s0[k] >= b0 && s1[k] >= b1 && s2[k] >= b2; any = any || in[k]; }
        if (any && pass()) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (in[k]) emit(s1[k], s2[k], (uint32_t)k);
        }
    }
}

// straight into the tile-major visibility buffer in HBM (no-return 64-bit min at the L2): the
// path of the triangles that cover a handful of pixels, which never enter a bin
// kEarlyZ: the fragment is first compared with the word as a load sees it and only goes on to the atomic when it
// would win.  The word only ever decreases, so a value read earlier can cause a redundant atomic, never a skipped one:
// the result is the same bits.  The load is an agent-scope atomic load (served by the L2, where the atomics execute;
// a line in the CU's own L1 could show the cleared word for as long as it stays there).  Frames with depth complexity
// (fragments per pixel far above one: C4's ten layers) run this variant — there nine of ten fragments lose, and a
// load is cheaper at the L2 than a 64-bit minimum; at one fragment per pixel the load is pure overhead (C3: +16 %),
// which is why it is a compile-time variant and not a branch.
template <bool kEarlyZ>
struct GlobalSinkT {
    unsigned long long* vis;
    uint32_t tiles_x, rank, world;
    uint32_t plane_stride;              // words between the planes of consecutive MSAA samples
    uint32_t test;                      // kEarlyZ: compare before the atomic (uniform; off in the first of two passes, whose surfaces hide nothing of each other: the load was a dependent round trip per raster round for nothing)
    __device__ __forceinline__ void operator()(int32_t i, int32_t j, float z, uint32_t id, uint32_t smp = 0) const {
        z = fminf(fmaxf(z, 0.0f), 1.0f);
        const uint32_t tx = (uint32_t)i / kTile, ty = (uint32_t)j / kTile;
        if (z < 1.0f && owns_tile(tx, ty, rank, world)) {
            unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | id;
            // a 32-bit byte offset from the (uniform) base: the largest target, 8192 x 8192 x 4 samples, is 2^31 bytes
            // (kMaxRenderSize), and the address is then one scalar base + one vector offset
            const uint32_t at = smp * plane_stride + (ty * tiles_x + tx) * kTilePixels + (((uint32_t)j % kTile) * kTile + ((uint32_t)i % kTile));
            unsigned long long* const word = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(vis) + (at << 3));
            if constexpr (kEarlyZ) {
                if (test && !(key < __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return;
            }
            __hip_atomic_fetch_min(word, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
};
typedef GlobalSinkT<false> GlobalSink;

template <bool kAlpha, int kSamples, class Sink>
__device__ __forceinline__ void raster_small(const SetupRecord& r, int32_t i0, int32_t i1, int32_t j0, int32_t j1,
                                             const Sink& sink, const FrameBuffers& fb, FrameConstRef fc) {
    // inside-positive edges: w0 = F(v2,v1,p), w1 = F(v0,v2,p), w2 = F(v1,v0,p)
    const int32_t ex0 = r.x1 - r.x2, ey0 = r.y1 - r.y2;
    const int32_t ex1 = r.x2 - r.x0, ey1 = r.y2 - r.y0;
    const int32_t ex2 = r.x0 - r.x1, ey2 = r.y0 - r.y1;
    const int32_t b0 = ((-ey0 > 0) || (ey0 == 0 && ex0 > 0)) ? 0 : 1;
    const int32_t b1 = ((-ey1 > 0) || (ey1 == 0 && ex1 > 0)) ? 0 : 1;
    const int32_t b2 = ((-ey2 > 0) || (ey2 == 0 && ex2 > 0)) ? 0 : 1;
    const int32_t px = i0 * kSub + kSub / 2, py = j0 * kSub + kSub / 2;
    // callers pass records with flag bit 0 (bbox below 64 px) and pixels of that bbox: every factor
    // is below 2^15 in magnitude, so the full-rate 24-bit multiply is exact
    int32_t r0 = mul24(ex0, py - r.y2) - mul24(ey0, px - r.x2);
    int32_t r1 = mul24(ex1, py - r.y0) - mul24(ey1, px - r.x0);
    int32_t r2 = mul24(ex2, py - r.y1) - mul24(ey2, px - r.x1);
    const SampleSteps<int32_t> st = sample_steps<int32_t>(ex0, ey0, ex1, ey1, ex2, ey2);
    for (int32_t j = j0; j <= j1; ++j) {
        int32_t w0 = r0, w1 = r1, w2 = r2;
        for (int32_t i = i0; i <= i1; ++i) {
            pixel_samples<kSamples, int32_t>(w0, w1, w2, b0, b1, b2, st,
                [&] { if constexpr (kAlpha) return !(r.flags & 2u) || alpha_pass(fb, fc, r.tri_id, (uint32_t)i, (uint32_t)j); return true; },
                [&](int32_t s1, int32_t s2, uint32_t smp) {
                    float l1 = (float)s1 * r.inv_area, l2 = (float)s2 * r.inv_area;
                    float z = (l1 * r.dz1 + l2 * r.dz2) + r.z0;
                    sink(i, j, z, r.tri_id, smp);
                });
            w0 -= ey0 * kSub; w1 -= ey1 * kSub; w2 -= ey2 * kSub;
        }
        r0 += ex0 * kSub; r1 += ex1 * kSub; r2 += ex2 * kSub;
    }
}

// Adds this wave's records to the per-tile counts.  Records that touch ONE tile (the bulk) are
// grouped by tile with ballots: one returning atomic per (wave, tile) group on the "single" count,
// and each lane's position inside the tile's bin is base + its rank in the group, so the later
// bin fill needs no atomics for them.  The atomics' results are only consumed after the grouping
// loop (one wait for all of them).  Records that touch several tiles are counted per tile in the
// "multi" counts and get their bin positions from a cursor in k_bin_fill.
// Must be called by all lanes of the wave that are still active (emit false = nothing to add).
// Returns the in-bin position for single-tile records, kNoBinPos otherwise.
constexpr uint32_t kNoBinPos = 0xFFFFFFFFu;

// Tile walk of multi-tile records.  A lane whose rect holds at most kWideTiles tiles walks it alone; the
// rects above that (a floor, a wall, a sky quad: up to every tile of the frame) are walked by the whole
// wave, one after the other, 64 tiles at a time: one lane looping over the 8160 tiles of a screen-filling
// triangle at 4K took 0.3 ms in the count pass and 2.8 ms (returning atomics) in the bin fill.
// Must be called in wave-uniform control flow; f(tx, ty, payload of the rect's lane).
constexpr uint32_t kWideTiles = 8;
template <class F>
__device__ __forceinline__ void walk_rects(bool has, uint32_t tx0, uint32_t ty0, uint32_t tx1, uint32_t ty1, uint32_t payload,
                                           uint32_t rank, uint32_t world, F&& f) {
    const uint32_t bw = tx1 - tx0 + 1u, bh = ty1 - ty0 + 1u;
    const bool wide = has && bw * bh > kWideTiles;
    if (has && !wide) {
        for (uint32_t ty = ty0; ty <= ty1; ++ty) {
            if (!owns_tile(tx0, ty, rank, world)) continue;
            for (uint32_t tx = tx0; tx <= tx1; ++tx) f(tx, ty, payload);
        }
    }
    const uint32_t lane = lane_id();
    for (unsigned long long m = __ballot(wide); m; m &= m - 1ull) {
        const int src = __ffsll((long long)m) - 1;
        const uint32_t x0 = __shfl(tx0, src), y0 = __shfl(ty0, src), w = __shfl(bw, src), h = __shfl(bh, src), pl = __shfl(payload, src);
        // rows of the rect are dealt to groups of lanes: lane -> (column, row step) without a division per tile
        const uint32_t cols = min(w, 64u), rows_per_pass = 64u / cols;
        const uint32_t lx = lane % cols, ly = lane / cols;
        if (ly < rows_per_pass)
            for (uint32_t ty = y0 + ly; ty < y0 + h; ty += rows_per_pass) {
                if (!owns_tile(x0, ty, rank, world)) continue;
                for (uint32_t tx = x0 + lx; tx < x0 + w; tx += cols) f(tx, ty, pl);
            }
    }
}

constexpr uint32_t kWidePos = 0xFFFFFFFEu;      // tileword.y of a record that is on the wide list instead of in bins

// A record that touches more than kWideListTiles tiles is not binned at all: its slot goes on a short
// list that every tile's raster workgroup checks against its own position.  (Binning one
// screen-filling triangle at 4K meant 8160 returning atomics by a single wave: 50 us.)  When the list
// is full the record is binned like any other.
__device__ __forceinline__ bool push_wide(bool candidate, const TileRect& t, uint32_t slot, FrameConstRef fc, FrameCounters* counters,
                                          uint32_t* __restrict__ wide_list, uint32_t cap_wide) {
    // every tile pays a look at every wide record, so only rects of at least an eighth of the frame's
    // tiles qualify (at 4K: a megapixel): with 64 tiles as the bar a soup of 300-pixel triangles put
    // thousands of records on the list and lost in the raster what it gained in the bin fill
    const uint32_t bar = max(kWideListTiles, (fc.tiles_x * fc.tiles_y) >> 3);
    if (!candidate || (t.tx1 - t.tx0 + 1u) * (t.ty1 - t.ty0 + 1u) <= bar) return false;
    const uint32_t w = atomicAdd(&counters->n_wide, 1u);
    if (w >= cap_wide) return false;
    wide_list[w] = slot;
    return true;
}

__device__ __forceinline__ uint32_t count_wave(bool emit, const TileRect& t, uint32_t slot, FrameConstRef fc,
                                               uint32_t* __restrict__ count_single,
                                               uint32_t* __restrict__ count_multi, FrameCounters* counters,
                                               uint32_t* __restrict__ wide_list, uint32_t cap_wide) {
    const uint32_t lane = lane_id();
    if (emit && !guard(t.tx1 < fc.tiles_x && t.ty1 < fc.tiles_y && t.tx0 <= t.tx1 && t.ty0 <= t.ty1, counters, kGuardCount)) emit = false;
    bool single = emit && t.tx0 == t.tx1 && t.ty0 == t.ty1;
    const uint32_t tile = t.ty0 * fc.tiles_x + t.tx0;
    if (single && !owns_tile(t.tx0, t.ty0, fc.rank, fc.world)) single = false, emit = false;
    uint32_t my_base = 0, my_leader = lane, my_rank = 0;
    unsigned long long pending = __ballot(single);
    while (pending) {
        int leader = __ffsll((long long)pending) - 1;
        uint32_t lt = __shfl(tile, leader);
        bool mine = single && tile == lt;
        unsigned long long same = __ballot(mine);
        if (lane == (uint32_t)leader) my_base = atomicAdd(&count_single[lt], (uint32_t)__popcll(same));
        if (mine) { my_leader = (uint32_t)leader; my_rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull)); }
        pending &= ~same;
    }
    uint32_t base = __shfl(my_base, my_leader);
    const uint32_t tiles_x = fc.tiles_x;
    const bool wide = push_wide(emit && !single, t, slot, fc, counters, wide_list, cap_wide);
    walk_rects(emit && !single && !wide, t.tx0, t.ty0, t.tx1, t.ty1, 0u, fc.rank, fc.world,
               [&](uint32_t tx, uint32_t ty, uint32_t) { atomicAdd(&count_multi[ty * tiles_x + tx], 1u); });
    return single ? base + my_rank : (wide ? kWidePos : kNoBinPos);
}

// Reject / cull / record of one triangle from its three pre-snapped vertices (K1 did the per-vertex
// work: snap_and_classify); sets need_clip (and emits nothing) when it must go through the clipper.
__device__ __forceinline__ void setup_triangle(bool live, const SnapVert& v0, const SnapVert& v1, const SnapVert& v2, uint32_t gid,
                                               FrameConstRef fc, int32_t j_lo, int32_t j_hi, SetupRecord& rec, TileRect& tr, bool& emit,
                                               bool& need_clip) {
    const uint32_t any = v0.code | v1.code | v2.code, all = v0.code & v1.code & v2.code;
    // dropped: a non-finite coordinate, or all three vertices outside one plane of the view volume
    const bool go = live && !(any & kOutNonFinite) && !(all & kOutPlanes);
    need_clip = go && (any & kOutNeedsClip) != 0u;
    // straight-line: the record is made for every lane (integers and one reciprocal: garbage in, defined garbage out)
    // and the verdict is a predicate; a branch around it made every field of rec and tr a merge of "computed" and
    // "zero" (some forty moves per 64 triangles)
    const Snapped s0{v0.x + fc.sub_dx, v0.y + fc.sub_dy, v0.z}, s1{v1.x + fc.sub_dx, v1.y + fc.sub_dy, v1.z},
                  s2{v2.x + fc.sub_dx, v2.y + fc.sub_dy, v2.z};
    const bool rec_ok = make_record(s0, s1, s2, gid, fc, j_lo, j_hi, rec, tr);
    emit = go && !need_clip && !(any & kOutBehind) && rec_ok;
    if (fc.world > 1u) emit = emit && any_owned(tr, fc);
}

// Direct raster, wave-cooperative.  Per wave only a quarter of the triangles survive culling, and a
// survivor covers a handful of pixel centres, so a per-lane pixel loop runs at the pace of the
// largest bbox with most lanes idle.  Instead the survivors are compacted into LDS (rank by ballot)
// and re-dealt four lanes per triangle: lane L of a pass takes triangle L/4 and its bbox pixels
// L%4, L%4+4, ...; 16 triangles per pass.
struct DirectTri {                  // 20 dwords = 5 x 16 B
    int32_t r0, r1, r2;             // edge functions at the bbox's first pixel centre
    int32_t sx0, sx1, sx2;          // ex * 256: step per pixel row
    int32_t nsy0, nsy1, nsy2;       // -ey * 256: step per pixel column
    uint32_t shape;                 // bbox width | pixel count << 8 | top-left bias bits << 16 | alpha test << 19
    float z0, dz1, dz2, inv_area;
    uint32_t id;
    uint32_t origin;                // i0 | j0 << 16
    float inv_w;                    // 1 / bbox width
    uint32_t smp;                   // MSAA, k_setup's direct path: the one sample this entry resolves (its plane); else 0
    uint32_t pad[2];
};
static_assert(sizeof(DirectTri) == 80, "DirectTri is 5 x 16 bytes");

// smp < 0: the edge values are those of the bbox's first pixel CENTRE; smp >= 0 (MSAA, k_setup's direct path): of
// MSAA sample `smp` of that pixel, and `t` is that sample's own pixel range
__device__ __forceinline__ DirectTri make_direct(const SetupRecord& r, const TileRect& t, int smp = -1) {
    // same integers as raster_small (bbox below 64 px: every factor fits 24 bits)
    const int32_t ex0 = r.x1 - r.x2, ey0 = r.y1 - r.y2;
    const int32_t ex1 = r.x2 - r.x0, ey1 = r.y2 - r.y0;
    const int32_t ex2 = r.x0 - r.x1, ey2 = r.y0 - r.y1;
    const uint32_t b0 = ((-ey0 > 0) || (ey0 == 0 && ex0 > 0)) ? 0u : 1u;
    const uint32_t b1 = ((-ey1 > 0) || (ey1 == 0 && ex1 > 0)) ? 0u : 1u;
    const uint32_t b2 = ((-ey2 > 0) || (ey2 == 0 && ex2 > 0)) ? 0u : 1u;
    const int32_t px = t.i0 * kSub + kSub / 2, py = t.j0 * kSub + kSub / 2;
    DirectTri d;
    d.r0 = mul24(ex0, py - r.y2) - mul24(ey0, px - r.x2);
    d.r1 = mul24(ex1, py - r.y0) - mul24(ey1, px - r.x0);
    d.r2 = mul24(ex2, py - r.y1) - mul24(ey2, px - r.x1);
    d.sx0 = ex0 * kSub; d.sx1 = ex1 * kSub; d.sx2 = ex2 * kSub;
    d.nsy0 = -ey0 * kSub; d.nsy1 = -ey1 * kSub; d.nsy2 = -ey2 * kSub;
    const uint32_t w = (uint32_t)(t.i1 - t.i0 + 1), h = (uint32_t)(t.j1 - t.j0 + 1);
    d.shape = w | ((w * h) << 8) | (b0 << 16) | (b1 << 17) | (b2 << 18) | (((r.flags >> 1) & 1u) << 19);
    d.z0 = r.z0; d.dz1 = r.dz1; d.dz2 = r.dz2; d.inv_area = r.inv_area;
    d.id = r.tri_id;
    d.origin = (uint32_t)t.i0 | ((uint32_t)t.j0 << 16);
    d.inv_w = __builtin_amdgcn_rcpf((float)w);
    d.smp = 0; d.pad[0] = d.pad[1] = 0;
    if (smp >= 0) {
        const SampleSteps<int32_t> st = sample_steps<int32_t>(ex0, ey0, ex1, ey1, ex2, ey2);
        const int32_t sg = (smp & 2) ? -1 : 1;                    // samples 2, 3 mirror samples 1, 0
        const bool use_a = smp == 0 || smp == 3;
        d.r0 += sg * (use_a ? st.a0 : st.b0); d.r1 += sg * (use_a ? st.a1 : st.b1); d.r2 += sg * (use_a ? st.a2 : st.b2);
        d.smp = (uint32_t)smp;
    }
    return d;
}

// pixel k (row-major in the bbox) of a compacted triangle
// kTriPlane: the triangle entry names the one sample plane it resolves (k_setup's MSAA rounds)
template <bool kAlpha, int kSamples, bool kTriPlane, class Sink>
__device__ __forceinline__ void direct_pixel(const DirectTri& d, uint32_t k, const Sink& sink, const FrameBuffers& fb,
                                             FrameConstRef fc) {
    const uint32_t w = d.shape & 255u;
    // k / w for k < 64, w <= 64: (k + 0.5) / w is at least 1/128 away from an integer, far beyond the
    // 1 ulp of the approximate reciprocal (absolute error below 1e-5 at these magnitudes)
    const uint32_t dj = (uint32_t)(((float)k + 0.5f) * d.inv_w);
    const uint32_t di = k - dj * w;
    const int32_t w0 = mad24((int32_t)di, d.nsy0, mad24((int32_t)dj, d.sx0, d.r0));
    const int32_t w1 = mad24((int32_t)di, d.nsy1, mad24((int32_t)dj, d.sx1, d.r1));
    const int32_t w2 = mad24((int32_t)di, d.nsy2, mad24((int32_t)dj, d.sx2, d.r2));
    const int32_t b0 = (int32_t)((d.shape >> 16) & 1u), b1 = (int32_t)((d.shape >> 17) & 1u), b2 = (int32_t)((d.shape >> 18) & 1u);
    const uint32_t i = (d.origin & 0xFFFFu) + di, j = (d.origin >> 16) + dj;
    SampleSteps<int32_t> st{};
    if constexpr (kSamples > 1)     // (sx = ex * 256, sy = ey * 256: the shifts are exact)
        st = sample_steps<int32_t>(d.sx0 >> kSubpixelBits, -(d.nsy0 >> kSubpixelBits), d.sx1 >> kSubpixelBits, -(d.nsy1 >> kSubpixelBits),
                          d.sx2 >> kSubpixelBits, -(d.nsy2 >> kSubpixelBits));
    pixel_samples<kSamples, int32_t>(w0, w1, w2, b0, b1, b2, st,
        [&] { if constexpr (kAlpha) return !((d.shape >> 19) & 1u) || alpha_pass(fb, fc, d.id, i, j); return true; },
        [&](int32_t s1, int32_t s2, uint32_t smp) {
            const float l1 = (float)s1 * d.inv_area, l2 = (float)s2 * d.inv_area;
            const float z = (l1 * d.dz1 + l2 * d.dz2) + d.z0;
            sink((int32_t)i, (int32_t)j, z, d.id, kTriPlane ? d.smp : smp);
        });
}

// A triangle whose bbox holds at most fb.direct_pixels pixel centres (kDirectPixels; the bulk at C2..C5 density) is
// rasterized right here by its lane, straight into the visibility buffer (GlobalSink); only the
// larger ones become records and go through the bins to the per-tile raster kernel.
// Slots of the record stream are deterministic: workgroup b owns slots [b*chunk_tris,
// (b+1)*chunk_tris) (FrameConst::chunk_tris, a multiple of 256), fills them densely from the front with a ballot + LDS prefix over its
// waves and leaves the number it used in chunk_count[b].  No global counter sits on this path (one
// word takes ~88 atomics/us on MI355X: 10M triangles / 64 lanes on one counter cost 1.4 ms here).

// One workgroup per surviving work unit (draw item, chunk of kChunkTris triangles): the item is
// uniform for the whole workgroup, so its fields come in through scalar loads.
// kAlpha: some draw item of the frame takes the per-fragment texture alpha test (compiled out otherwise)
// kSamples: 1, or 4 (MSAA): every candidate pixel is tested at its four sample positions, one visibility plane each
// kViews: the pass has several views (a light's shadow cascades), the unit names its view; the single-view
// instantiation is the frame's main pass with the view's terms folded away
// kEarlyZ: the direct path's fragments are compared with the visibility word before the atomic (GlobalSinkT)
template <bool kAlpha, int kSamples, bool kViews, bool kEarlyZ>
__global__ __launch_bounds__(256) void k_setup(FrameBuffers fb, Capacities cap) {
    FrameConstRef fc = frame_const_ref(fb.frame_const);
    __shared__ uint32_t s_used;         // record slots of this workgroup's chunk handed out so far
    __shared__ uint32_t s_valid[4], s_frags[4];
    __shared__ uint4 s_direct[4][64][5];   // per wave: the compacted small triangles of one iteration
    // Workgroups go to the eight XCDs round-robin (workgroup b to XCD b % 8), and each XCD has its own L2.  Units
    // next to each other in the list share vertices (the neighbouring rows of a mesh lie a few chunks apart), so
    // every XCD takes one contiguous eighth of the list instead of every eighth unit.
    // (early-depth frames in two passes: the first pass takes the units k_cull listed — the near items' —, the second those
    // k_cull_far appended behind them)
    const uint32_t setup_pass = kEarlyZ ? fb.setup_pass : 0u;
    uint32_t range_lo = 0u, range_n = fb.counters->n_active;
    if constexpr (kEarlyZ) {
        if (setup_pass == 2u) { range_lo = min(fb.counters->n_near, range_n); range_n -= range_lo; }
    }
    const uint32_t per_xcd = (range_n + kXcds - 1u) / kXcds;
    const uint32_t range_index = (blockIdx.x % kXcds) * per_xcd + blockIdx.x / kXcds;
    if (blockIdx.x / kXcds >= per_xcd || range_index >= range_n) return;
    const uint32_t unit_index = range_lo + range_index;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint2 unit = fb.active_units[unit_index];
    const uint32_t item_s = __builtin_amdgcn_readfirstlane(unit.x), chunk_s = __builtin_amdgcn_readfirstlane(unit.y) & ((1u << kUnitViewShift) - 1u);
    const uint32_t view_s = kViews ? __builtin_amdgcn_readfirstlane(unit.y) >> kUnitViewShift : 0u;
    const int32_t j_lo = (int32_t)(view_s * fc.view_h), j_hi = kViews ? j_lo + (int32_t)fc.view_h - 1 : (int32_t)fc.height - 1;       // the view's rows
    const SnapVert* const snap_v = fb.snap + (size_t)view_s * fc.n_verts;
    if (!guard(item_s < fc.n_items, fb.counters, kGuardSetup)) return;      // uniform over the workgroup
    if (threadIdx.x == 0) s_used = 0;
    __syncthreads();
    typedef const __attribute__((address_space(4))) DrawItem* cptr_item;
    cptr_item d = (cptr_item)fb.items + item_s;
    // Two passes in frames with depth complexity (kEarlyZ, setup_pass 1 and 2): the units of the nearest draw items (the
    // host orders the items near to far) in pass 1, then k_hiz takes the farthest depth of every 8x8 pixel block, k_cull_far
    // keeps the far units that can still show, k_transform runs for their vertices, and pass 2 drops a small triangle whose
    // nearest vertex lies behind everything in the blocks of its bbox before any pixel work.  Pass 0: every unit at once,
    // per-fragment depth test only.
    const uint32_t chunk_tris = fc.chunk_tris;
    const uint32_t first_t = chunk_s * chunk_tris;                         // first triangle of the chunk within its item
    const uint32_t item_tris = d->n_tris;
    const uint32_t n_here = item_tris > first_t ? min(item_tris - first_t, chunk_tris) : 0u;
    const uint32_t vb = d->vert_base, nv = d->n_verts, tri_base = d->tri_base;
    const bool alpha_item = (d->flags & kItemAlphaTest) != 0u;       // fragments take the texture alpha test
    gptr_u32 indices = (gptr_u32)d->indices;
    const bool tables_ok = guard(vb + nv <= fc.n_verts, fb.counters, kGuardSetup);
    const uint32_t chunk_base = unit_index * chunk_tris;
    const uint32_t n_tiles = fc.tiles_x * fc.tiles_y;
    const GlobalSinkT<kEarlyZ> direct_sink{reinterpret_cast<unsigned long long*>(fb.vis), fc.tiles_x, fc.rank, fc.world, (uint32_t)fb.vis_stride, setup_pass != 1u ? 1u : 0u};
    uint32_t valid = 0, frags = 0;      // (frags: per lane, summed at the end)
    // (a unit without triangles or with tables that do not fit runs no iteration; the loads below need nv >= 1)
    const uint32_t n_iters = (tables_ok && nv != 0u) ? (n_here + 255u) / 256u : 0u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // RND_SETUP_PIPE 1: the next iteration's index triple is requested before this iteration's vertices (C5 setup 0.686 ->
    // 0.654 ms, C3 0.103 -> 0.101); 2: its vertices too (91 VGPRs, five waves per SIMD instead of seven: C5 0.723, C3 0.110)
#ifndef RND_SETUP_PIPE
#define RND_SETUP_PIPE 1
#endif
    // straight-line: a lane beyond the chunk's end works on the chunk's first triangle, an index beyond the
    // item's vertices is clamped and reported; both are dropped by the predicate
    auto index_ptr = [&](uint32_t iter) {
        const uint32_t local = iter * 256u + threadIdx.x;
        return indices + 3ull * (first_t + (local < n_here ? local : 0u));
    };
#if RND_SETUP_PIPE >= 1
    uint32_t ni0 = 0, ni1 = 0, ni2 = 0;         // the next iteration's index triple, in flight while this one is worked on
    if (n_iters) { gptr_u32 ix = index_ptr(0); ni0 = ix[0]; ni1 = ix[1]; ni2 = ix[2]; }
#endif
#if RND_SETUP_PIPE >= 2
    u32x4 nq0 = {}, nq1 = {}, nq2 = {};          // ... and its three snapped vertices
    bool n_ok = false;
    if (n_iters) {
        n_ok = max(ni0, max(ni1, ni2)) < nv;
        nq0 = *reinterpret_cast<const u32x4*>(snap_v + vb + min(ni0, nv - 1u));
        nq1 = *reinterpret_cast<const u32x4*>(snap_v + vb + min(ni1, nv - 1u));
        nq2 = *reinterpret_cast<const u32x4*>(snap_v + vb + min(ni2, nv - 1u));
        if (n_iters > 1u) { gptr_u32 ix = index_ptr(1); ni0 = ix[0]; ni1 = ix[1]; ni2 = ix[2]; }
    }
#endif
    for (uint32_t iter = 0; iter < n_iters; ++iter) {
        const uint32_t local = iter * 256u + threadIdx.x;
        const uint32_t gid = tri_base + first_t + local;                   // draw-order triangle id
        bool emit = false, need_clip = false;
        float z_near = 0.0f;
        SetupRecord rec;
        TileRect tr;
        {
            const bool in = local < n_here;
#if RND_SETUP_PIPE >= 2
            u32x4 q0 = nq0, q1 = nq1, q2 = nq2;
            const bool idx_ok = n_ok;
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));
            if (iter + 1u < n_iters) {
                n_ok = max(ni0, max(ni1, ni2)) < nv;
                nq0 = *reinterpret_cast<const u32x4*>(snap_v + vb + min(ni0, nv - 1u));
                nq1 = *reinterpret_cast<const u32x4*>(snap_v + vb + min(ni1, nv - 1u));
                nq2 = *reinterpret_cast<const u32x4*>(snap_v + vb + min(ni2, nv - 1u));
                if (iter + 2u < n_iters) { gptr_u32 ix = index_ptr(iter + 2u); ni0 = ix[0]; ni1 = ix[1]; ni2 = ix[2]; }
            }
            guard(idx_ok, fb.counters, kGuardSetup);
#else
#if RND_SETUP_PIPE == 1
            uint32_t i0 = ni0, i1 = ni1, i2 = ni2;
            if (iter + 1u < n_iters) { gptr_u32 ix = index_ptr(iter + 1u); ni0 = ix[0]; ni1 = ix[1]; ni2 = ix[2]; }
#else
            gptr_u32 ix = index_ptr(iter);
            uint32_t i0 = ix[0], i1 = ix[1], i2 = ix[2];
#endif
            const bool idx_ok = max(i0, max(i1, i2)) < nv;
            guard(idx_ok, fb.counters, kGuardSetup);
            i0 = min(i0, nv - 1u); i1 = min(i1, nv - 1u); i2 = min(i2, nv - 1u);
            u32x4 q0 = *reinterpret_cast<const u32x4*>(snap_v + vb + i0);
            u32x4 q1 = *reinterpret_cast<const u32x4*>(snap_v + vb + i1);
            u32x4 q2 = *reinterpret_cast<const u32x4*>(snap_v + vb + i2);
            // three whole 16-byte gathers in one round trip: left alone the compiler loads the outcodes
            // first and sinks the positions into the branch that uses them (six gathers, two round trips)
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));
#endif
            const SnapVert v0{(int32_t)q0.x, (int32_t)q0.y, __uint_as_float(q0.z), q0.w};
            const SnapVert v1{(int32_t)q1.x, (int32_t)q1.y, __uint_as_float(q1.z), q1.w};
            const SnapVert v2{(int32_t)q2.x, (int32_t)q2.y, __uint_as_float(q2.z), q2.w};
            {
                // a wave whose 64 triangles are all rejected, back-facing or degenerate (the far side of a closed mesh is
                // long runs of them) stops here: the same outcode and area terms as setup_triangle / make_record
                const uint32_t any = v0.code | v1.code | v2.code, all = v0.code & v1.code & v2.code;
                const bool go = in && idx_ok && !(any & kOutNonFinite) && !(all & kOutPlanes);
                const long long A2 = (long long)(v1.x - v0.x) * (v2.y - v0.y) - (long long)(v2.x - v0.x) * (v1.y - v0.y);
                if (!__ballot(go && ((any & kOutNeedsClip) != 0u || A2 < 0))) continue;
            }
            setup_triangle(in && idx_ok, v0, v1, v2, gid, fc, j_lo, j_hi, rec, tr, emit, need_clip);
            if (alpha_item) rec.flags |= 2u;
            if constexpr (kEarlyZ) z_near = fminf(v0.z, fminf(v1.z, v2.z));
        }
        // queue the triangles that need the clipper (wave-compacted, rare)
        unsigned long long cm = __ballot(need_clip);
        if (cm) {
            if (lane == 0) fb.counters->chain_needed = 1u;       // (frames launched without k_clip .. k_raster are withheld and re-run: k_shade's gate)
            int leader = __ffsll((long long)cm) - 1;
            uint32_t base = 0;
            if (lane == (uint32_t)leader) base = atomicAdd(&fb.counters->n_clip, (uint32_t)__popcll(cm));
            base = __shfl(base, leader);
            if (need_clip) {
                uint32_t q = base + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull));
                if (q < cap.clip_queue) { fb.clip_queue[q] = gid; fb.clip_queue[cap.clip_queue + q] = view_s; } else atomicOr(&fb.counters->overflow, 2u);
            }
        }
        valid += (uint32_t)__popcll(__ballot(emit));
        // small triangles: rasterize now, no record
        const uint32_t box_px = (uint32_t)(tr.i1 - tr.i0 + 1) * (uint32_t)(tr.j1 - tr.j0 + 1);
        bool direct = emit && (rec.flags & 1u) && box_px <= fb.direct_pixels;
#ifndef RND_NO_FRAG_COUNT
        // (pixel centres in the bboxes of the small triangles, whatever becomes of them: the host's measure of depth complexity)
        frags += direct ? box_px : 0u;
#endif
        if constexpr (kEarlyZ) {
            if (setup_pass == 2u) {
                // Occluded as a whole?  The depth of every sample of the triangle is at least its nearest vertex's up to the
                // rounding of the plane's evaluation (below 1e-6); if that, less a margin, is still behind the farthest depth
                // k_hiz found in every 8x8 block the bbox touches (depths only decrease afterwards), no sample can win.
                const uint32_t bx0 = (uint32_t)tr.i0 >> 3, bx1 = (uint32_t)tr.i1 >> 3, by0 = (uint32_t)tr.j0 >> 3, by1 = (uint32_t)tr.j1 >> 3;
                bool occluded = direct && (bx1 - bx0 + 1u) * (by1 - by0 + 1u) <= 4u;      // (a bbox of at most 32 pixels spans more only as a thin strip)
                if (occluded) {
                    float farthest = 0.0f;
                    for (uint32_t by = by0; by <= by1; ++by)
                        for (uint32_t bx = bx0; bx <= bx1; ++bx)
                            farthest = fmaxf(farthest, fb.hiz[((by >> 2) * fc.tiles_x + (bx >> 2)) * 16u + ((by & 3u) * 4u + (bx & 3u))]);
                    occluded = z_near - 2.0e-6f > farthest;
                }
                if (occluded) { direct = false; emit = false; }
            }
        }
        // The wave's direct triangles are compacted into LDS and re-dealt 2, 4 or 8 lanes each, by the size of the
        // wave's bboxes: a pass costs its entry fetch plus one round per ceil(pixels / lanes) of its largest bbox, so
        // sub-pixel triangles (C3, C5: one or two pixel centres per bbox) want many triangles per pass and ~20-pixel
        // bboxes (C4) many lanes per triangle.  Measured with the width fixed at 2 / 4 / 8: k_setup at C3 0.102 / 0.107 /
        // 0.137 ms, C2 0.038 / 0.034 / 0.034, C4 3.44 / 2.51 / 2.40, C5 0.667 / 0.694 / 0.785.
        auto run_direct = [&](bool has, const TileRect& t, int smp) {
            const unsigned long long dm = __ballot(has);
            if (!dm) return;
            const uint32_t box = (uint32_t)(t.i1 - t.i0 + 1) * (uint32_t)(t.j1 - t.j0 + 1);
            const uint32_t n_all = (uint32_t)__popcll(dm), n_3 = (uint32_t)__popcll(__ballot(has && box >= 3u)),
                           n_13 = (uint32_t)__popcll(__ballot(has && box >= 13u));
            const uint32_t sh = n_13 * 2u > n_all ? 3u : (n_3 * 3u > n_all ? 2u : 1u);       // log2(lanes per triangle), uniform
            if (has) {
                const DirectTri dt = make_direct(rec, t, smp);
                const uint4* src = reinterpret_cast<const uint4*>(&dt);
                uint4* dst = s_direct[wave][__popcll(dm & ((1ull << lane) - 1ull))];
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3]; dst[4] = src[4];
            }
            // the wave's own LDS traffic is ordered by the hardware; keep the compiler from moving the
            // reads above the writes (and the next round's writes above these reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t n_direct = (uint32_t)__popcll(dm);
            for (uint32_t base = 0; base < n_direct; base += 64u >> sh) {
                const uint32_t which = base + (lane >> sh);
                if (which < n_direct) {
                    DirectTri dt;
                    uint4* dst = reinterpret_cast<uint4*>(&dt);
                    const uint4* src = s_direct[wave][which];
                    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3]; dst[4] = src[4];
                    const uint32_t npix = (dt.shape >> 8) & 255u;
                    for (uint32_t k = lane & ((1u << sh) - 1u); k < npix; k += 1u << sh) direct_pixel<kAlpha, 1, (kSamples > 1)>(dt, k, direct_sink, fb, fc);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        if constexpr (kSamples == 1) {
            run_direct(direct, tr, -1);
        } else {
            // MSAA: one round per sample, each over the pixels whose sample `smp` lies in the bbox (a sub-pixel
            // triangle has ~4 such pixels per sample, against 9 pixels x 4 samples when walked pixel by pixel);
            // the triangle's set-up is shared by the four rounds
            const int32_t xmin = min(rec.x0, min(rec.x1, rec.x2)), xmax = max(rec.x0, max(rec.x1, rec.x2));
            const int32_t ymin = min(rec.y0, min(rec.y1, rec.y2)), ymax = max(rec.y0, max(rec.y1, rec.y2));
#pragma unroll
            for (int smp = 0; smp < 4; ++smp) {
                constexpr int32_t kDx[4] = {-32, 96, -96, 32}, kDy[4] = {-96, -32, 32, 96};
                TileRect ts = tr;
                ts.i0 = max((xmin - kDx[smp] - kSub / 2 + kSub - 1) >> kSubpixelBits, 0);
                ts.i1 = min((xmax - kDx[smp] - kSub / 2) >> kSubpixelBits, (int32_t)fc.width - 1);
                ts.j0 = max((ymin - kDy[smp] - kSub / 2 + kSub - 1) >> kSubpixelBits, j_lo);
                ts.j1 = min((ymax - kDy[smp] - kSub / 2) >> kSubpixelBits, j_hi);
                run_direct(direct && ts.i0 <= ts.i1 && ts.j0 <= ts.j1, ts, smp);
            }
        }
        if (direct) emit = false;
        // the rest become records: the wave takes its slots from the workgroup's chunk with one LDS
        // atomic (the waves of a workgroup do not wait for each other)
        const unsigned long long em = __ballot(emit);
        if (em == 0ull) continue;
        uint32_t wave_base = 0;
        if (lane == 0) { fb.counters->chain_needed = 1u; wave_base = atomicAdd(&s_used, (uint32_t)__popcll(em)); }
        wave_base = __builtin_amdgcn_readfirstlane(wave_base);
        const uint32_t slot = chunk_base + wave_base + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
        const uint32_t pos = count_wave(emit, tr, slot, fc, fb.tile_count, fb.tile_count + n_tiles, fb.counters, fb.wide_list, cap.wide_list);
        if (emit && guard(slot < cap.records, fb.counters, kGuardSetup)) {
            float4* dst = reinterpret_cast<float4*>(fb.records + slot);
            const float4* src = reinterpret_cast<const float4*>(&rec);
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
            fb.tilewords[slot] = make_uint2(pack_tileword(tr), pos);
        }
    }
    // bbox pixels of the direct path's triangles (the host reads depth complexity off their sum: FrameReport::n_fragments)
    for (int o = 32; o > 0; o >>= 1) frags += __shfl_xor(frags, o);
    if (lane == 0) { s_valid[wave] = valid; s_frags[wave] = frags; }
    __syncthreads();
    if (threadIdx.x == 0) {
        fb.chunk_count[unit_index] = s_used;
        // (striped: one word would take ten to a hundred thousand atomics per frame, at ~88 per microsecond)
        const uint32_t v = s_valid[0] + s_valid[1] + s_valid[2] + s_valid[3], f = s_frags[0] + s_frags[1] + s_frags[2] + s_frags[3];
        if (v | f) atomicAdd(&fb.stat_stripes[(unit_index % kStatStripes) * kStatStride], (unsigned long long)v | ((unsigned long long)f << 32));
    }
}

// Farthest depth of every 8x8 pixel block of the frame, from the visibility words as they stand between the two
// passes of k_setup's early-depth variant: 16 floats per 32x32 tile (block (bx, by) of the tile at by * 4 + bx).
// One workgroup per tile of the frame; a tile this context does not own (sort-first) reads as 1.0: nothing is
// dropped against it.  Depth bits of [0, 1] order as unsigned integers.
__global__ __launch_bounds__(256) void k_hiz(FrameBuffers fb) {
    FrameConstRef fc = frame_const_ref(fb.frame_const);
    __shared__ uint32_t s_part[4][16];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tx = tile % fc.tiles_x, ty = tile / fc.tiles_x;
    if (tile == 0u && tid == 0u) fb.counters->n_near = fb.counters->n_active;        // (the first pass's units: k_cull_far appends behind them)
    const uint32_t n_tiles = fc.tiles_x * fc.tiles_y;          // (behind the blocks' 16 per tile: one value per tile, their maximum)
    if (!owns_tile(tx, ty, fc.rank, fc.world)) {
        if (tid < 16u) fb.hiz[tile * 16u + tid] = 1.0f;
        if (tid == 0u) fb.hiz[n_tiles * 16u + tile] = 1.0f;
        return;
    }
    const unsigned long long* vis = reinterpret_cast<const unsigned long long*>(fb.vis) + (size_t)tile * kTilePixels;
    // thread tid holds column tid % 32 of rows tid / 32 + 8 k: block column (tid % 32) / 8 of block row k
    uint32_t m[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) m[k] = (uint32_t)(vis[k * 256u + tid] >> 32);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        // the wave holds two rows of 32 columns: the 8 columns of a block (lanes xor 1, 2, 4), then the two rows (xor 32)
        for (int o : {1, 2, 4, 32}) m[k] = max(m[k], (uint32_t)__shfl_xor((int)m[k], o));
    }
    if ((lane & 39u) == 0u) {            // lanes 0, 8, 16, 24: one per block column
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) s_part[wave][k * 4u + (lane >> 3)] = m[k];
    }
    __syncthreads();
    uint32_t far_bits = 0u;
    if (tid < 16u) {
        far_bits = max(max(s_part[0][tid], s_part[1][tid]), max(s_part[2][tid], s_part[3][tid]));
        fb.hiz[tile * 16u + tid] = __uint_as_float(far_bits);
    }
    if (wave == 0u) {
        for (int o : {1, 2, 4, 8}) far_bits = max(far_bits, (uint32_t)__shfl_xor((int)far_bits, o));
        if (tid == 0u) fb.hiz[n_tiles * 16u + tile] = __uint_as_float(far_bits);
    }
}

// The far items' units of an early-depth frame, after the near items are drawn: a unit whose bounds (k_cull: the screen rect
// of its box's corners and a depth nothing of it is nearer than) lie behind k_hiz's farthest depth in every 8x8 block of the
// rect cannot show (depths only decrease afterwards) and goes no further: its vertices are not transformed, its triangles
// not set up.  The others join the active list and flag their vertex blocks (the second row of flags).  One candidate per
// wave, the wave's lanes over the blocks of the rect; one atomic per workgroup of four.
__global__ __launch_bounds__(256) void k_cull_far(FrameBuffers fb) {
    FrameConstRef fc = frame_const_ref(fb.frame_const);
    __shared__ uint32_t s_keep[4];
    const uint32_t n_far = min(fb.counters->n_far, fc.n_units);
    const uint32_t first = blockIdx.x * 4u;
    if (first >= n_far) return;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    {
        const uint32_t c = first + wave;        // (uniform over the wave)
        bool keep = false;
        if (c < n_far) {
            keep = true;
            const uint4 ub = fb.unit_bounds[c];
            const uint32_t bx0 = ub.x & 0xFFFFu, by0 = ub.x >> 16, bw = (ub.y & 0xFFFFu) - bx0 + 1u, bh = (ub.y >> 16) - by0 + 1u;
            const uint32_t n_b = bw * bh;
            if (ub.w != 0u && n_b <= 16384u) {
                const float z_near_unit = __uint_as_float(ub.z);
                // whole tiles first (k_hiz's second level: a sixteenth of the loads; at C4 it settles nearly every unit)
                const uint32_t tx0 = bx0 >> 2, ty0 = by0 >> 2, tw = ((bx0 + bw - 1u) >> 2) - tx0 + 1u, th = ((by0 + bh - 1u) >> 2) - ty0 + 1u;
                const float* const hiz_tiles = fb.hiz + (size_t)fc.tiles_x * fc.tiles_y * 16u;
                const float inv_tw = __builtin_amdgcn_rcpf((float)tw);
                bool open = false;
                for (uint32_t k = lane; k < tw * th; k += 64u) {
                    uint32_t dy = (uint32_t)(((float)k + 0.5f) * inv_tw);
                    dy = dy * tw > k ? dy - 1u : ((dy + 1u) * tw <= k ? dy + 1u : dy);        // (exact k / tw from the estimate)
                    open = open || !(z_near_unit > hiz_tiles[(ty0 + dy) * fc.tiles_x + tx0 + (k - dy * tw)]);
                }
                if (__ballot(open) != 0ull) {
                    const float inv_bw = __builtin_amdgcn_rcpf((float)bw);
                    open = false;
                    for (uint32_t k = lane; k < n_b; k += 64u) {
                        uint32_t dy = (uint32_t)(((float)k + 0.5f) * inv_bw);
                        dy = dy * bw > k ? dy - 1u : ((dy + 1u) * bw <= k ? dy + 1u : dy);
                        const uint32_t bx = bx0 + (k - dy * bw), by = by0 + dy;
                        const float farthest = fb.hiz[((by >> 2) * fc.tiles_x + (bx >> 2)) * 16u + ((by & 3u) * 4u + (bx & 3u))];
                        open = open || !(z_near_unit > farthest);
                    }
                }
                keep = __ballot(open) != 0ull;
            }
        }
        if (lane == 0u) s_keep[wave] = keep ? 1u : 0u;
    }
    __syncthreads();
    if (wave != 0u) return;
    const uint32_t c = first + lane;
    const bool mine = lane < 4u && c < n_far;
    const bool keep = mine && s_keep[lane & 3u] != 0u;
    const unsigned long long km = __ballot(keep);
    uint32_t base = 0;
    if (km) {
        const int leader = __ffsll((long long)km) - 1;
        if (lane == (uint32_t)leader) base = atomicAdd(&fb.counters->n_active, (uint32_t)__popcll(km));
        base = __shfl(base, leader);
    }
    if (mine) {
        const uint2 u = fb.far_units[c];
        const DrawItem& d = fb.items[u.x];
        if (keep) {
            const uint32_t n_vblocks = (fc.n_verts + kVertexBlock - 1) / kVertexBlock;
            fb.active_units[base + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = make_uint2(u.x, u.y);
            const ChunkInfo ci = d.chunks[u.y];
            const uint32_t vb0 = (d.vert_base + ci.vmin) / kVertexBlock, vb1 = (d.vert_base + ci.vmax) / kVertexBlock;
            if (vb1 - vb0 >= 64u) fb.counters->all_verts = 1u;
            else for (uint32_t b = vb0; b <= vb1 && b < n_vblocks; ++b) fb.vblock_needed[n_vblocks + b] = u.x + 1u;
        } else {
            // (the host's measure of depth complexity keeps counting what was turned away, at the rate of the last plain
            // frame, or the variant would switch itself off)
            const uint32_t first_t = u.y * fc.chunk_tris;
            const uint32_t n_here = d.n_tris > first_t ? min(d.n_tris - first_t, fc.chunk_tris) : 0u;
            const unsigned long long f = ((unsigned long long)n_here * fb.bbox_px_hint_x16) >> 4;
            if (f) atomicAdd(&fb.stat_stripes[(c % kStatStripes) * kStatStride], f << 32);
        }
    }
}

// --- clipper (rare path): near, far and the four guard-band planes, then a fan -------------
struct ClipVert { float x, y, z, w; };

__device__ __forceinline__ float plane_dist(const ClipVert& v, int plane) {
    switch (plane) {
        case 0: return v.z;
        case 1: return v.w - v.z;
        case 2: return v.x + kGuardBand * v.w;
        case 3: return kGuardBand * v.w - v.x;
        case 4: return v.y + kGuardBand * v.w;
        default: return kGuardBand * v.w - v.y;
    }
}

// evaluated from the inside vertex towards the outside one (watertight across shared edges)
__device__ __forceinline__ ClipVert clip_intersect(const ClipVert& in, float din, const ClipVert& out, float dout) {
    float t = din / (din - dout);
    ClipVert r;
    r.x = in.x + t * (out.x - in.x);
    r.y = in.y + t * (out.y - in.y);
    r.z = in.z + t * (out.z - in.z);
    r.w = in.w + t * (out.w - in.w);
    return r;
}

__global__ __launch_bounds__(256) void k_clip(FrameBuffers fb, Capacities cap) {
    FrameConstRef fc = frame_const_ref(fb.frame_const);
    const uint32_t n = min(fb.counters->n_clip, cap.clip_queue);
    uint32_t* const count_multi = fb.tile_count + fc.tiles_x * fc.tiles_y;
    const uint32_t tiles_x = fc.tiles_x;
    // wave-uniform loops (the tile walk of a wide record is shared by the wave): a lane without a
    // triangle, or whose polygon is gone, just has an empty fan
    for (uint32_t q0 = blockIdx.x * 256u; q0 < n; q0 += gridDim.x * 256u) {
        const uint32_t q = q0 + threadIdx.x;
        Snapped s[12];
        int fan = 0;                        // triangles of this lane's fan
        uint32_t gid = 0, alpha = 0, view = 0;
        if (q < n) {
            gid = fb.clip_queue[q];
            view = min(fb.clip_queue[cap.clip_queue + q], kMaxViews - 1u);
            uint32_t it = find_item(fb.tri_bases, fc.n_items, gid);
            const DrawItem& d = fb.items[it];
            uint32_t t = gid - d.tri_base;
            gptr_u32 ix = (gptr_u32)d.indices + 3ull * t;
            alpha = (d.flags & kItemAlphaTest) ? 2u : 0u;
            ClipVert a[12], b[12];
            // the clip positions again from the object-space positions, by K1's own routine (the same operations on the
            // same inputs: the same bits).  Triangles that reach the clipper are rare, and this way the depth-only
            // passes (a light's four cascade views) do not store 16 bytes per vertex and view that nothing else reads.
            for (int k = 0; k < 3; ++k) {
                const uint32_t v = min(ix[k], max(d.n_verts, 1u) - 1u);
                const float4 c = clip_position(d, view_pv(fc, view), ((gptr_f32)d.px)[v], ((gptr_f32)d.py)[v], ((gptr_f32)d.pz)[v]);
                a[k] = ClipVert{c.x, c.y, c.z, c.w};
            }
            int cnt = 3;
            for (int plane = 0; plane < 6 && cnt >= 3; ++plane) {
                int m = 0;
                for (int k = 0; k < cnt; ++k) {
                    const ClipVert& p = a[k];
                    const ClipVert& nx = a[(k + 1 == cnt) ? 0 : k + 1];
                    float dp = plane_dist(p, plane), dn = plane_dist(nx, plane);
                    bool ip = dp >= 0.0f, in = dn >= 0.0f;
                    if (ip) b[m++] = p;
                    if (ip != in) b[m++] = ip ? clip_intersect(p, dp, nx, dn) : clip_intersect(nx, dn, p, dp);
                }
                cnt = m;
                for (int k = 0; k < cnt; ++k) a[k] = b[k];
            }
            if (cnt >= 3) {
                float half_w = 0.5f * (float)fc.width, half_h = 0.5f * (float)fc.view_h;
                bool ok = true;
                const int32_t view_dy = (int32_t)(view * fc.view_h) * kSub;
                for (int k = 0; k < cnt; ++k) { bool o = snap_vertex(a[k].x, a[k].y, a[k].z, a[k].w, half_w, half_h, s[k], fc.sub_dx, fc.sub_dy + view_dy); ok = ok && o; }
                if (ok) fan = cnt - 2;
            }
        }
        for (int k = 1; __ballot(k <= fan) != 0ull; ++k) {
            bool emit = k <= fan;
            SetupRecord rec;
            TileRect tr{0, 0, 0, 0, 0, 0, 0, 0};
            if (emit) emit = make_record(s[0], s[k], s[k + 1], gid, fc, (int32_t)(view * fc.view_h), (int32_t)((view + 1u) * fc.view_h) - 1, rec, tr) && any_owned(tr, fc);
            if (emit) {
                rec.flags |= alpha;
                const uint32_t slot = cap.setup_slots + atomicAdd(&fb.counters->n_records, 1u);
                if (slot >= cap.records) { atomicOr(&fb.counters->overflow, 1u); emit = false; }
                else {
                    atomicAdd(&fb.counters->n_valid, 1u);
                    fb.records[slot] = rec;
                    const bool wide = push_wide(true, tr, slot, fc, fb.counters, fb.wide_list, cap.wide_list);
                    fb.tilewords[slot] = make_uint2(pack_tileword(tr), wide ? kWidePos : kNoBinPos);
                    if (wide) emit = false;      // nothing to count
                }
            }
            walk_rects(emit, tr.tx0, tr.ty0, tr.tx1, tr.ty1, 0u, fc.rank, fc.world,
                       [&](uint32_t tx, uint32_t ty, uint32_t) { atomicAdd(&count_multi[ty * tiles_x + tx], 1u); });
        }
    }
}

// --- exclusive scan of the tile counts (one workgroup) --------------------------------------
__global__ __launch_bounds__(1024) void k_scan_tiles(FrameBuffers fb, Capacities cap) {
    FrameConstRef fc = frame_const_ref(fb.frame_const);
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_valid[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_tiles = fc.tiles_x * fc.tiles_y;
    // statistics: triangles that survived setup (per-unit counts of k_setup; the clipper adds its own).
    // These loads are issued first so they overlap the scan's.
    const uint32_t n_active = min(fb.counters->n_active, fc.n_units);
    const uint32_t n_wide = min(fb.counters->n_wide, cap.wide_list);
    uint32_t v = 0, used = 0;
    unsigned long long fr = 0;          // bbox pixels of the direct path's triangles (depth complexity, for the host)
    if (tid < kStatStripes) { const unsigned long long w = fb.stat_stripes[tid * kStatStride]; v = (uint32_t)w; fr = w >> 32; }
    for (uint32_t c0 = 0; c0 < n_active; c0 += 8u * 1024u) {
        uint32_t slots[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {          // independent loads in flight, not a chain of round trips
            const uint32_t c = c0 + k * 1024u + tid;
            slots[k] = c < n_active ? fb.chunk_count[c] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) used |= slots[k];
    }
    // A pass in which no unit wrote a record and the clipper wrote none either (every triangle took k_setup's direct path:
    // C3, C5, the depth-only passes of such scenes) has no tile counts to scan: k_bin_fill and k_raster look at
    // n_entries and n_wide alone (both zero then) before anything else, so the offsets, cursors and slice counts of an
    // earlier frame may stay as they are.  Saves the tile loop's three barriers and two round trips in a kernel that is
    // one workgroup on the frame's critical chain.
    __shared__ uint32_t s_any;
    if (tid == 0) s_any = fb.counters->n_records | n_wide;
    __syncthreads();
    if (used) atomicOr(&s_any, 1u);
    __syncthreads();
    const bool any_records = s_any != 0u;
    // Tiles go through LDS in batches of 8192 so that global traffic is coalesced (this is a single
    // workgroup on one CU: with a strided pattern its one texture pipe was the whole kernel time).
    // Per batch: coalesced load of count sums into LDS, every thread scans its run of 8 consecutive
    // tiles, offsets go back through LDS and out coalesced.
    constexpr uint32_t kBatch = 8192, kRun = kBatch / 1024;
    __shared__ uint32_t s_tiles[kBatch + kBatch / 32];          // padded: a run of 8 starts in its own bank group
    __shared__ uint32_t s_batch_total;
    uint32_t carry = 0;
    for (uint32_t batch = 0; any_records && batch < n_tiles; batch += kBatch) {
#pragma unroll
        for (uint32_t k = 0; k < kRun; ++k) {
            const uint32_t local = k * 1024u + tid, i = batch + local;
            s_tiles[local + local / 32u] = i < n_tiles ? fb.tile_count[i] + fb.tile_count[n_tiles + i] : 0u;
        }
        __syncthreads();
        uint32_t cnt[kRun], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < kRun; ++k) { const uint32_t local = tid * kRun + k; cnt[k] = s_tiles[local + local / 32u]; sum += cnt[k]; }
        // heavy tiles (more than 256 candidates: bin entries + the wide list) are shared by several raster
        // workgroups: the first is the tile's own, the others are work items (tile, slice) on a short list
#pragma unroll
        for (uint32_t k = 0; k < kRun; ++k) {
            const uint32_t t = batch + tid * kRun + k;
            if (t >= n_tiles) continue;
            const uint32_t n_all = cnt[k] + n_wide;
            uint32_t slices = 1;
            if (n_all > 256u && owns_tile(t % fc.tiles_x, t / fc.tiles_x, fc.rank, fc.world)) {
                const uint32_t want = min((n_all + 255u) / 256u, kRasterMaxSlices);
                for (; slices < want; ++slices) {
                    const uint32_t at = atomicAdd(&fb.counters->n_extra, 1u);
                    if (at >= kRasterExtraCap) break;
                    fb.raster_extra[at] = make_uint2(t, slices);
                }
                fb.tile_slices[t] = slices;     // (only read for tiles with more than 256 candidates)
            }
        }
        uint32_t inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t up = __shfl_up(inc, o);
            if (lane >= (uint32_t)o) inc += up;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wave_off = 0;
        for (uint32_t w = 0; w < wave; ++w) wave_off += s_wave[w];
        uint32_t run = carry + wave_off + inc - sum;
#pragma unroll
        for (uint32_t k = 0; k < kRun; ++k) { const uint32_t local = tid * kRun + k; s_tiles[local + local / 32u] = run; run += cnt[k]; }
        if (tid == 1023) s_batch_total = run;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kRun; ++k) {
            const uint32_t local = k * 1024u + tid, i = batch + local;
            if (i < n_tiles) { fb.tile_offset[i] = s_tiles[local + local / 32u]; fb.tile_cursor[i] = 0; }
        }
        carry = s_batch_total;
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); fr += __shfl_xor(fr, o); }
    __shared__ unsigned long long s_frags[16];
    if (lane == 0) { s_valid[wave] = v; s_frags[wave] = fr; }
    __syncthreads();
    if (tid == 1023) {
        const uint32_t n_entries = carry;
        uint32_t total = 0;
        unsigned long long frags_total = 0;
        for (uint32_t w = 0; w < 16; ++w) { total += s_valid[w]; frags_total += s_frags[w]; }
        atomicAdd(&fb.counters->n_valid, total);
        fb.counters->n_fragments = (uint32_t)min(frags_total, 0xFFFFFFFFull);
        fb.counters->n_entries = n_entries;
        if (n_entries > cap.entries) atomicOr(&fb.counters->overflow, 4u);
    }
}

// --- K2b: scatter record slots into the bins ---------------------------------------------
// Wave-uniform: every lane of the wave calls it, `has` says whether the lane holds record r.
__device__ __forceinline__ void bin_record(const FrameBuffers& fb, FrameConstRef fc, const Capacities& cap, bool has, uint32_t r) {
    uint2 w = make_uint2(0u, 0u);
    if (has) w = fb.tilewords[r];
    const uint32_t tx0 = w.x & 255u, ty0 = (w.x >> 8) & 255u;
    if (has && !guard(tx0 + ((w.x >> 16) & 255u) < fc.tiles_x && ty0 + (w.x >> 24) < fc.tiles_y, fb.counters, kGuardBin)) has = false;
    if (has && w.y == kWidePos) has = false;        // on the wide list: no bin entries
    if (has && w.y != kNoBinPos) {
        // single-tile record: its position was handed out by the count pass
        const uint32_t tile = ty0 * fc.tiles_x + tx0;
        const uint32_t idx = fb.tile_offset[tile] + w.y;
        if (idx < cap.entries) fb.entries[idx] = r;
    }
    const uint32_t tiles_x = fc.tiles_x, n_entries = cap.entries;
    walk_rects(has && w.y == kNoBinPos, tx0, ty0, tx0 + ((w.x >> 16) & 255u), ty0 + (w.x >> 24), r, fc.rank, fc.world,
               [&](uint32_t tx, uint32_t ty, uint32_t rec) {
                   const uint32_t tile = ty * tiles_x + tx;
                   const uint32_t idx = fb.tile_offset[tile] + fb.tile_count[tile] + atomicAdd(&fb.tile_cursor[tile], 1u);
                   if (idx < n_entries) fb.entries[idx] = rec;
               });
}

// Workgroup b < n_chunks takes the records k_setup's workgroup b left at the front of its chunk;
// the remaining workgroups stride over the clipper's records behind the deterministic slots.
constexpr uint32_t kBinClipBlocks = 64;

__global__ __launch_bounds__(256) void k_bin_fill(FrameBuffers fb, Capacities cap) {
    FrameConstRef fc = frame_const_ref(fb.frame_const);
    const uint32_t n_chunks = fc.n_units;
    if (fb.counters->n_entries == 0u) return;        // nothing was counted into any bin
    if (blockIdx.x < n_chunks) {
        if (blockIdx.x >= fb.counters->n_active) return;
        const uint32_t base = blockIdx.x * fc.chunk_tris;
        const uint32_t n = min(fb.chunk_count[blockIdx.x], fc.chunk_tris);
        for (uint32_t k0 = 0; k0 < n; k0 += 256u) bin_record(fb, fc, cap, k0 + threadIdx.x < n, base + k0 + threadIdx.x);
    } else {
        const uint32_t n = min(cap.setup_slots + fb.counters->n_records, cap.records);
        for (uint32_t r0 = cap.setup_slots + (blockIdx.x - n_chunks) * 256u; r0 < n; r0 += kBinClipBlocks * 256u)
            bin_record(fb, fc, cap, r0 + threadIdx.x < n, r0 + threadIdx.x);
    }
}

// ------------------------------------------------------------------------------------------
// K3: per-tile raster.  One 256-thread workgroup per owned 32x32 tile; the tile's
// (depth bits << 32 | triangle id) words live in LDS and fragments resolve with ds_min_u64, so the
// nearest fragment wins and equal depths go to the first triangle in draw order, whatever order
// the bin is processed in.  Small triangles (the bulk at C2..C5 density) are rasterized by one
// lane each with 32-bit edge functions; the rest are queued in LDS and swept by a whole wave in
// 8x8 pixel blocks with 64-bit edge functions.
// ------------------------------------------------------------------------------------------
// kNarrow: the record's bbox is below 64 px (flag bit 0), so every factor of the edge functions fits the
// full-rate 24-bit multiply and 32-bit sums (as in raster_small); otherwise 64-bit arithmetic.  Medium
// triangles (tens of pixels, the bulk of real content) are narrow: with 64-bit edges this sweep was
// 200 wave instructions per bin entry and made k_raster VALU-bound (C2 at 8K).
template <bool kAlpha, bool kNarrow, int kSamples>
__device__ __forceinline__ void raster_wave(const SetupRecord& r, int32_t i0, int32_t i1, int32_t j0, int32_t j1,
                                            const LdsSink& sink, const FrameBuffers& fb, FrameConstRef fc) {
    typedef typename std::conditional<kNarrow, int32_t, long long>::type edge_t;
    const uint32_t lane = lane_id();
    const int32_t lx = (int32_t)(lane & 7u), ly = (int32_t)(lane >> 3);
    const int32_t ex0 = r.x1 - r.x2, ey0 = r.y1 - r.y2;
    const int32_t ex1 = r.x2 - r.x0, ey1 = r.y2 - r.y0;
    const int32_t ex2 = r.x0 - r.x1, ey2 = r.y0 - r.y1;
    const edge_t b0 = ((-ey0 > 0) || (ey0 == 0 && ex0 > 0)) ? 0 : 1;
    const edge_t b1 = ((-ey1 > 0) || (ey1 == 0 && ex1 > 0)) ? 0 : 1;
    const edge_t b2 = ((-ey2 > 0) || (ey2 == 0 && ex2 > 0)) ? 0 : 1;
    // The lane's edge functions at its pixel of the first block; from block to block they move by whole
    // multiples of the coefficients (exact integer arithmetic: the same values as evaluating them anew,
    // which cost seven instructions per edge and block instead of one).  Lanes past i1 / j1 run up to 7
    // pixels outside the bbox: for narrow records that still fits 32 bits (factors below 2^15 + 2^11).
    const int32_t px = (i0 + lx) * kSub + kSub / 2, py = (j0 + ly) * kSub + kSub / 2;
    edge_t r0, r1, r2;
    if constexpr (kNarrow) {
        r0 = mul24(ex0, py - r.y2) - mul24(ey0, px - r.x2);
        r1 = mul24(ex1, py - r.y0) - mul24(ey1, px - r.x0);
        r2 = mul24(ex2, py - r.y1) - mul24(ey2, px - r.x1);
    } else {
        r0 = (long long)ex0 * (py - r.y2) - (long long)ey0 * (px - r.x2);
        r1 = (long long)ex1 * (py - r.y0) - (long long)ey1 * (px - r.x0);
        r2 = (long long)ex2 * (py - r.y1) - (long long)ey2 * (px - r.x1);
    }
    const SampleSteps<edge_t> st = sample_steps<edge_t>(ex0, ey0, ex1, ey1, ex2, ey2);
    const edge_t sx0 = (edge_t)ey0 * (8 * kSub), sx1 = (edge_t)ey1 * (8 * kSub), sx2 = (edge_t)ey2 * (8 * kSub);   // per block column (subtracted)
    const edge_t sy0 = (edge_t)ex0 * (8 * kSub), sy1 = (edge_t)ex1 * (8 * kSub), sy2 = (edge_t)ex2 * (8 * kSub);   // per block row
    for (int32_t by = j0; by <= j1; by += 8) {
        edge_t w0 = r0, w1 = r1, w2 = r2;
        const int32_t j = by + ly;
        for (int32_t bx = i0; bx <= i1; bx += 8) {
            const int32_t i = bx + lx;
            if (i <= i1 && j <= j1)
                pixel_samples<kSamples, edge_t>(w0, w1, w2, b0, b1, b2, st,
                    [&] { if constexpr (kAlpha) return !(r.flags & 2u) || alpha_pass(fb, fc, r.tri_id, (uint32_t)i, (uint32_t)j); return true; },
                    [&](edge_t s1, edge_t s2, uint32_t smp) {
                        float l1 = (float)s1 * r.inv_area, l2 = (float)s2 * r.inv_area;
                        float z = (l1 * r.dz1 + l2 * r.dz2) + r.z0;
                        sink(i, j, z, r.tri_id, smp);
                    });
            w0 -= sx0; w1 -= sx1; w2 -= sx2;
        }
        r0 += sy0; r1 += sy1; r2 += sy2;
    }
}

__device__ __forceinline__ SetupRecord load_record(const SetupRecord* __restrict__ recs, uint32_t slot) {
    const float4* src = reinterpret_cast<const float4*>(recs + slot);
    SetupRecord r;
    float4* dst = reinterpret_cast<float4*>(&r);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    return r;
}

template <bool kAlpha, int kSamples>
__global__ __launch_bounds__(256) void k_raster(FrameBuffers fb, Capacities cap) {
    FrameConstRef fc = frame_const_ref(fb.frame_const);
    __shared__ unsigned long long s_vis[kTilePixels * kSamples];        // one plane per MSAA sample
    // the batch's triangles that a whole wave sweeps, kept in LDS by the lanes that classified them: a
    // second fetch of the record per triangle from global memory made the sweep a queue of ~1 us round
    // trips (C2 at 8K, ~37 bin entries of ~70 px per tile: 13 us per tile, 0.41 ms per frame)
    __shared__ float4 s_big[256][3];
    // two classes share s_big: "medium" triangles (narrow, at most 255 pixel centres inside this tile) fill it
    // from the front and are dealt kMediumLanes lanes each with the direct-raster pixel walk (exact bbox, no
    // 8x8 granularity, no per-block edge set-up); the rest fill it from the back and are swept by a whole wave
    __shared__ uint32_t s_nmed, s_nlarge;
#ifndef RND_MEDIUM_LANES
#define RND_MEDIUM_LANES 16
#endif
    constexpr uint32_t kMediumLanes = RND_MEDIUM_LANES;
    const uint32_t tid = threadIdx.x;
    // a pass without bin entries and without wide records (C3: every triangle took the direct path in k_setup) has
    // nothing here: one scalar load decides it for all ten thousand workgroups
    if (fb.counters->n_entries == 0u && fb.counters->n_wide == 0u) return;
    // workgroups [0, n_owned) are the tiles' own; the ones behind them take the (tile, slice) work items
    // k_scan_tiles listed for heavy tiles, whose candidates are dealt to the slices batch by batch
    uint32_t tile, slice = 0;
    if (blockIdx.x < fb.n_owned) tile = fb.launch_order[blockIdx.x].tile;
    else {
        const uint32_t at = blockIdx.x - fb.n_owned;
        if (at >= min(fb.counters->n_extra, kRasterExtraCap)) return;
        const uint2 item = fb.raster_extra[at];
        tile = item.x; slice = item.y;
        if (!guard(tile < fc.tiles_x * fc.tiles_y, fb.counters, kGuardRaster)) return;
    }
    const int32_t tile_x0 = (int32_t)((tile % fc.tiles_x) * kTile), tile_y0 = (int32_t)((tile / fc.tiles_x) * kTile);
    const int32_t tile_x1 = min(tile_x0 + (int32_t)kTile, (int32_t)fc.width) - 1;
    const int32_t tile_y1 = min(tile_y0 + (int32_t)kTile, (int32_t)fc.height) - 1;
    uint32_t off = fb.tile_offset[tile], n = fb.tile_count[tile] + fb.tile_count[fc.tiles_x * fc.tiles_y + tile];
    if (off >= cap.entries) n = 0; else n = min(n, cap.entries - off);
    const uint32_t n_wide = min(fb.counters->n_wide, cap.wide_list);
    if (n == 0 && n_wide == 0) return;     // nothing binned here: the tile keeps what k_setup's direct path wrote
    const uint32_t my_tx = tile % fc.tiles_x, my_ty = tile / fc.tiles_x;
    const uint32_t n_all = n + n_wide;     // bin entries, then the wide list (filtered by its records' tile rects)
    const uint32_t n_slices = slice == 0u && n_all <= 256u ? 1u : max(fb.tile_slices[tile], 1u);
    if (slice * 256u >= n_all || slice >= n_slices) return;
    const bool shared_tile = n_slices > 1u;  // several workgroups on this tile: they meet in the global words by atomic min
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(fb.vis) + (size_t)tile * kTilePixels;
    constexpr int32_t pad = kSamples > 1 ? kSamplePad : 0;      // candidate pixels: any sample position inside the bbox
#pragma unroll
    for (uint32_t smp = 0; smp < (uint32_t)kSamples; ++smp)
        for (uint32_t k = tid; k < kTilePixels; k += 256u) s_vis[smp * kTilePixels + k] = slice == 0u ? dst[(size_t)smp * fb.vis_stride + k] : kClearKey;
    const LdsSink sink{s_vis, tile_x0, tile_y0};
    __syncthreads();
    for (uint32_t b = slice * 256u; b < n_all; b += n_slices * 256u) {
        if (tid == 0) { s_nmed = 0; s_nlarge = 0; }
        __syncthreads();
        uint32_t e = b + tid;
        bool mine = e < n_all;
        uint32_t slot = 0;
        if (mine) {
            slot = e < n ? fb.entries[off + e] : fb.wide_list[e - n];
            if (!guard(slot < cap.records, fb.counters, kGuardRaster)) slot = 0;
            if (e >= n) {
                const uint32_t tw = fb.tilewords[slot].x;
                const uint32_t rx0 = tw & 255u, ry0 = (tw >> 8) & 255u;
                mine = my_tx >= rx0 && my_tx <= rx0 + ((tw >> 16) & 255u) && my_ty >= ry0 && my_ty <= ry0 + (tw >> 24);
            }
        }
        if (mine) {
            SetupRecord r = load_record(fb.records, slot);
            int32_t xmin = min(r.x0, min(r.x1, r.x2)), xmax = max(r.x0, max(r.x1, r.x2));
            int32_t ymin = min(r.y0, min(r.y1, r.y2)), ymax = max(r.y0, max(r.y1, r.y2));
            int32_t i0 = max((xmin - pad - kSub / 2 + kSub - 1) >> kSubpixelBits, tile_x0);
            int32_t i1 = min((xmax + pad - kSub / 2) >> kSubpixelBits, tile_x1);
            int32_t j0 = max((ymin - pad - kSub / 2 + kSub - 1) >> kSubpixelBits, tile_y0);
            int32_t j1 = min((ymax + pad - kSub / 2) >> kSubpixelBits, tile_y1);
            if (i0 <= i1 && j0 <= j1) {
                uint32_t npix = (uint32_t)(i1 - i0 + 1) * (uint32_t)(j1 - j0 + 1);
                if ((r.flags & 1u) && npix <= kSmallPixels) raster_small<kAlpha, kSamples>(r, i0, i1, j0, j1, sink, fb, fc);
                else {
                    const bool medium = (r.flags & 1u) && npix <= 255u;
                    const uint32_t k = medium ? atomicAdd(&s_nmed, 1u) : 255u - atomicAdd(&s_nlarge, 1u);
                    const float4* src = reinterpret_cast<const float4*>(&r);
                    s_big[k][0] = src[0]; s_big[k][1] = src[1]; s_big[k][2] = src[2];
                }
            }
        }
        __syncthreads();
        const uint32_t nmed = s_nmed, nlarge = s_nlarge;
        for (uint32_t q = tid / kMediumLanes; q < nmed; q += 256u / kMediumLanes) {
            SetupRecord r;
            {
                float4* dst = reinterpret_cast<float4*>(&r);
                dst[0] = s_big[q][0]; dst[1] = s_big[q][1]; dst[2] = s_big[q][2];
            }
            int32_t xmin = min(r.x0, min(r.x1, r.x2)), xmax = max(r.x0, max(r.x1, r.x2));
            int32_t ymin = min(r.y0, min(r.y1, r.y2)), ymax = max(r.y0, max(r.y1, r.y2));
            TileRect tr{0, 0, 0, 0, 0, 0, 0, 0};
            tr.i0 = max((xmin - pad - kSub / 2 + kSub - 1) >> kSubpixelBits, tile_x0);
            tr.i1 = min((xmax + pad - kSub / 2) >> kSubpixelBits, tile_x1);
            tr.j0 = max((ymin - pad - kSub / 2 + kSub - 1) >> kSubpixelBits, tile_y0);
            tr.j1 = min((ymax + pad - kSub / 2) >> kSubpixelBits, tile_y1);
            const DirectTri dt = make_direct(r, tr);
            const uint32_t npix = (dt.shape >> 8) & 255u;
            for (uint32_t k = tid % kMediumLanes; k < npix; k += kMediumLanes) direct_pixel<kAlpha, kSamples, false>(dt, k, sink, fb, fc);
        }
        for (uint32_t q = tid >> 6; q < nlarge; q += 4u) {
            SetupRecord r;
            {
                float4* dst = reinterpret_cast<float4*>(&r);
                const uint32_t at = 255u - q;
                dst[0] = s_big[at][0]; dst[1] = s_big[at][1]; dst[2] = s_big[at][2];
            }
            int32_t xmin = min(r.x0, min(r.x1, r.x2)), xmax = max(r.x0, max(r.x1, r.x2));
            int32_t ymin = min(r.y0, min(r.y1, r.y2)), ymax = max(r.y0, max(r.y1, r.y2));
            int32_t i0 = max((xmin - pad - kSub / 2 + kSub - 1) >> kSubpixelBits, tile_x0);
            int32_t i1 = min((xmax + pad - kSub / 2) >> kSubpixelBits, tile_x1);
            int32_t j0 = max((ymin - pad - kSub / 2 + kSub - 1) >> kSubpixelBits, tile_y0);
            int32_t j1 = min((ymax + pad - kSub / 2) >> kSubpixelBits, tile_y1);
            if (r.flags & 1u) raster_wave<kAlpha, true, kSamples>(r, i0, i1, j0, j1, sink, fb, fc);
            else raster_wave<kAlpha, false, kSamples>(r, i0, i1, j0, j1, sink, fb, fc);
        }
        __syncthreads();
    }
    if (!shared_tile) {
#pragma unroll
        for (uint32_t smp = 0; smp < (uint32_t)kSamples; ++smp)
            for (uint32_t k = tid; k < kTilePixels; k += 256u) dst[(size_t)smp * fb.vis_stride + k] = s_vis[smp * kTilePixels + k];
    } else {
#pragma unroll
        for (uint32_t smp = 0; smp < (uint32_t)kSamples; ++smp)
            for (uint32_t k = tid; k < kTilePixels; k += 256u) {
                const unsigned long long v = s_vis[smp * kTilePixels + k];
                if (v != kClearKey) __hip_atomic_fetch_min(&dst[(size_t)smp * fb.vis_stride + k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
}

// ------------------------------------------------------------------------------------------
// K4: deferred shading.  One workgroup per owned tile, four pixels per lane.  Per pixel: the
// winning triangle's three clip positions give perspective-correct barycentrics (2D homogeneous
// edge functions, f64), the source vertices are gathered from the 32-byte AoS the reference
// uploads, and the ambient + per-light Blinn-Phong sum is evaluated once in f32, rounded once to
// f16 (the Rgba16Float target), tone-mapped and sRGB-encoded.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float round_f16(float x) { return __half2float(__float2half_rn(x)); }

__device__ __forceinline__ uint32_t srgb8(float lin) {
    float x = fminf(fmaxf(lin, 0.0f), 1.0f);
    float y = (x <= 0.0031308f) ? 12.92f * x : 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
    float q = y * 255.0f + 0.5f;
    int v = (int)q;
    return (uint32_t)min(max(v, 0), 255);
}

// Colour parity is a tolerance (<= 1/255 on >= 99.9 % of pixels), so the light loop uses the
// hardware reciprocal / reciprocal-square-root / exp2 / log2 (v_rcp_f32, v_rsq_f32: 1 ulp) instead
// of the correctly rounded sequences; build with -DRND_FAST_SHADE=0 for the bit-exact variant.
#ifndef RND_FAST_SHADE
#define RND_FAST_SHADE 1
#endif
__device__ __forceinline__ float rcp_f(float x) {
#if RND_FAST_SHADE
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
__device__ __forceinline__ float rsqrt_f(float x) {
#if RND_FAST_SHADE
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
__device__ __forceinline__ uint32_t srgb8_f(float lin) {
#if RND_FAST_SHADE
    // floor(255 y + 0.5) with the OETF's two branches folded into one FMA each: 255 * (1.055 p - 0.055) + 0.5 and
    // 255 * 12.92 x + 0.5.  x is clamped to [0, 1] (NaN -> 0), so the result is in [0, 255] without a second clamp.
    const float x = fminf(fmaxf(lin, 0.0f), 1.0f);
    const float p = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.4f));
    const float v = (x <= 0.0031308f) ? fmaf(x, 255.0f * 12.92f, 0.5f) : fmaf(p, 255.0f * 1.055f, 0.5f - 255.0f * 0.055f);
    return (uint32_t)(int)v;
#else
    return srgb8(lin);
#endif
}

// Every lane must be active; the results are wave-uniform.
// The wave's bounding box: minimum of lo[0..2] and maximum of hi[0..2] over the 64 lanes.  Four rotations inside
// the rows of 16 lanes, each ONE instruction per value (v_min_f32 / v_max_f32 with the DPP operand), then the four
// rows through readlane.  Written as assembly: from C++ a step is three instructions (a move with the DPP control,
// which the compiler does not fold into a float min because it first canonicalizes both operands against
// signalling NaNs), and round 1's detour through order-preserving integers added conversions both ways: 26 VALU
// instructions per value against 10.  A VGPR written by a VALU instruction must not be read through DPP for two
// wait states: the six chains are interleaved, so each value is read five instructions after it was written; the
// leading s_nop 1 covers whatever wrote the inputs last.
// Inputs are never NaN here (fragment positions that passed the raster, or the +-3e38 initial values).
__device__ __forceinline__ void wave_box(float (&lo)[3], float (&hi)[3]) {
    asm("s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %4, %4, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %5, %5, %5 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %4, %4, %4 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %5, %5, %5 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %5, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]));
    auto rows = [](float v, bool is_min) {
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
        return is_min ? fminf(fminf(r0, r1), fminf(r2, r3)) : fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = rows(lo[c], true); hi[c] = rows(hi[c], false); }
}

struct PixelInputs {
    float fx, fy, fz;       // world position (frag_pos)
    float Nx, Ny, Nz;       // normalized normal
    float Vx, Vy, Vz;       // normalized fragment -> camera
    float br, bg, bb;       // base colour
    float cr, cg, cb;       // accumulated colour
    bool lit;               // takes part in the light loop
};

// vs_main outputs re-derived for the winning triangle of one pixel, and the ambient term.
// `d` is wave-uniform (the caller waterfalls over the items of the wave), so its fields come in
// through scalar loads; per pixel only the index triple, the three clip positions and the corners' attributes are
// gathered.  Three stages, so that a lane with several pixels of one draw item can have the same stage of all of
// them in flight at once (the fragment phase is a chain of dependent round trips — visibility word, item, index
// triple, vertices — and its throughput follows the number of chains in flight: DESIGN.md 5):
//   frag_indices  issues the index-triple load
//   frag_fetch    clamps / checks the indices and issues the clip-position and attribute gathers
//   frag_finish   barycentrics, interpolation, texture, ambient term
struct FragFetch {
    uint32_t i0, i1, i2;
    f32x4 q0, q1, q2;               // clip positions K1 stored for the corners (frames that keep them)
    float p0[3], p1[3], p2[3];      // object-space positions (frames without stored clip positions: see corner_clip)
    float n0[3], n1[3], n2[3];      // normals
    float u0, v0, u1, v1, u2, v2;   // uv (textured items)
};

// The fragment's world position is needed for the lights.  Wherever the clip position IS the projection of that
// world position (every item but the FixedSize billboards, render_mesh.wgsl:50-52, :65-67) it comes back out of the
// interpolated clip x, y, w — under perspective_lh the view-space point is (x / P00, y / P11, w): no depth, no
// cancellation — through fc.unproject (camera_transform's linear part times diag(1/P00, 1/P11, 1)).  The clip
// positions are gathered anyway for the barycentrics, so the object-space positions are not: a corner is then the
// 16-byte clip position + the 12-byte normal (+ 8 bytes of uv under a texture) instead of the clip position + the
// whole 32-byte vertex in two 16-byte gathers, and the loaded data of a pixel is 21 registers instead of 36.
__device__ __forceinline__ bool world_from_clip(const DrawItem& d, FrameConstRef fc) {
    return fc.unproject_ok != 0u && (d.flags & kItemBillboardMask) != 2u;      // uniform over the wave
}

// The clip position of a corner (x, y, w: depth is not needed at shade time).  Two sources, chosen per frame on the
// host (VisualServer::enqueue_frame, `store_clip`):
//   * K1 stored it (fb.clip): a second divergent 16-byte gather per corner next to the 12-byte normal.  Costs K1 16 of
//     its 44 bytes per vertex; right when the frame has few vertices per pixel (C3: K1 +7 us, k_shade -18 us).
//   * made again from the object-space position that the corner's 32-byte vertex brings along with its normal, by K1's
//     own routine, operation by operation (clip_position: the bits setup and raster saw), 36 instructions per corner.
//     Right when vertices outnumber the pixels (C5: 50 M vertices, transform 0.36 -> 0.25 ms; no buffer of 16 B per
//     vertex and frame parity either).
// Both give the same bits, so the choice never shows in the image.
struct ClipXYW { float x, y, w; };
__device__ __forceinline__ ClipXYW corner_clip(const DrawItem& d, FrameConstRef fc, const float (&p)[3]) {
    const float4 c = clip_position(d, fc.pv, p[0], p[1], p[2]);
    return ClipXYW{c.x, c.y, c.w};
}

__device__ __forceinline__ void frag_indices(const DrawItem& d, uint32_t id, FragFetch& f, bool& ok) {
    // straight-line: an index that would leave its buffer is clamped and reported, not branched on
    uint32_t t = id - d.tri_base;
    ok = t < d.n_tris;
    t = ok ? t : 0u;
    // (12 bytes per triangle: 32-bit offsets reach 357 M triangles of one item; the host refuses larger submeshes)
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    const u32x3 ix = load_at<u32x3>(d.indices, t * 12u);
    f.i0 = ix.x; f.i1 = ix.y; f.i2 = ix.z;
}

template <bool kTextures, bool kStored>
__device__ __forceinline__ void frag_fetch(const DrawItem& d, FrameConstRef fc, const float4* __restrict__ clip, FragFetch& f,
                                           bool ok, FrameCounters* counters) {
    const uint32_t last_v = max(d.n_verts, 1u) - 1u;
    ok = ok && max(f.i0, max(f.i1, f.i2)) <= last_v && d.n_verts != 0u && d.vert_base + d.n_verts <= fc.n_verts;
    const uint32_t i0 = min(f.i0, last_v), i1 = min(f.i1, last_v), i2 = min(f.i2, last_v);
    guard(ok, counters, kGuardShade);
    const bool want_uv = kTextures && d.texture != RND_NONE;
    if (kStored && world_from_clip(d, fc)) {           // (every branch here is uniform over the wave)
        // (32 bytes per vertex: 32-bit offsets reach 134 M vertices of one item)
        typedef float f32x3 __attribute__((ext_vector_type(3)));
        const f32x3 a = load_at<f32x3>(d.verts, i0 * 32u + 12u), b = load_at<f32x3>(d.verts, i1 * 32u + 12u), c = load_at<f32x3>(d.verts, i2 * 32u + 12u);
        f.n0[0] = a.x; f.n0[1] = a.y; f.n0[2] = a.z;
        f.n1[0] = b.x; f.n1[1] = b.y; f.n1[2] = b.z;
        f.n2[0] = c.x; f.n2[1] = c.y; f.n2[2] = c.z;
        if (want_uv) {
            const f32x2 ua = load_at<f32x2>(d.verts, i0 * 32u + 24u), ub = load_at<f32x2>(d.verts, i1 * 32u + 24u), uc = load_at<f32x2>(d.verts, i2 * 32u + 24u);
            f.u0 = ua.x; f.v0 = ua.y; f.u1 = ub.x; f.v1 = ub.y; f.u2 = uc.x; f.v2 = uc.y;
        }
    } else {
        // a corner is its 32-byte source vertex: position + normal.x in the first 16 bytes, normal.yz (+ uv) in the
        // second half of the same 32-byte sector
        const f32x4 a0 = load_at<f32x4>(d.verts, i0 * 32u), b0 = load_at<f32x4>(d.verts, i1 * 32u), c0 = load_at<f32x4>(d.verts, i2 * 32u);
        f.p0[0] = a0.x; f.p0[1] = a0.y; f.p0[2] = a0.z; f.n0[0] = a0.w;
        f.p1[0] = b0.x; f.p1[1] = b0.y; f.p1[2] = b0.z; f.n1[0] = b0.w;
        f.p2[0] = c0.x; f.p2[1] = c0.y; f.p2[2] = c0.z; f.n2[0] = c0.w;
        if (want_uv) {
            const f32x4 a1 = load_at<f32x4>(d.verts, i0 * 32u + 16u), b1 = load_at<f32x4>(d.verts, i1 * 32u + 16u), c1 = load_at<f32x4>(d.verts, i2 * 32u + 16u);
            f.n0[1] = a1.x; f.n0[2] = a1.y; f.n1[1] = b1.x; f.n1[2] = b1.y; f.n2[1] = c1.x; f.n2[2] = c1.y;
            f.u0 = a1.z; f.v0 = a1.w; f.u1 = b1.z; f.v1 = b1.w; f.u2 = c1.z; f.v2 = c1.w;
        } else {
            const f32x2 a1 = load_at<f32x2>(d.verts, i0 * 32u + 16u), b1 = load_at<f32x2>(d.verts, i1 * 32u + 16u), c1 = load_at<f32x2>(d.verts, i2 * 32u + 16u);
            f.n0[1] = a1.x; f.n0[2] = a1.y; f.n1[1] = b1.x; f.n1[2] = b1.y; f.n2[1] = c1.x; f.n2[2] = c1.y;
        }
    }
    if constexpr (kStored) {
        // whole 16-byte loads: z is not used here, and left alone the compiler narrows each load to an 8-byte
        // and a 4-byte one — for divergent gathers the cost is per (lane, instruction), not per byte
        const uint32_t last_c = max(fc.n_verts, 1u) - 1u;
        // (16 bytes per vertex of the frame: the host keeps clip positions only in frames of fewer than 2^28 vertices)
        f.q0 = load_at<f32x4>(clip, min(d.vert_base + i0, last_c) * 16u); f.q1 = load_at<f32x4>(clip, min(d.vert_base + i1, last_c) * 16u);
        f.q2 = load_at<f32x4>(clip, min(d.vert_base + i2, last_c) * 16u);
    }
}

template <bool kTextures, bool kStored>
__device__ __forceinline__ void frag_finish(const DrawItem& d, FrameConstRef fc, const float4* __restrict__ clip,
                                            const TextureDesc* __restrict__ textures, const float* lut, uint32_t i, uint32_t j, FragFetch& f, PixelInputs& o) {
#if RND_FAST_SHADE
    // products and sums of the interpolation and of the two transforms may contract to FMAs here, as in the light loop:
    // colour parity is a tolerance (the coverage / depth arithmetic of the raster kernels never contracts)
#pragma clang fp contract(fast)
#endif
    const bool from_clip = world_from_clip(d, fc);
    ClipXYW c0, c1, c2;
    if constexpr (kStored) {
        asm volatile("" : "+v"(f.q0), "+v"(f.q1), "+v"(f.q2));
        c0 = ClipXYW{f.q0.x, f.q0.y, f.q0.w}; c1 = ClipXYW{f.q1.x, f.q1.y, f.q1.w}; c2 = ClipXYW{f.q2.x, f.q2.y, f.q2.w};
    } else {
        c0 = corner_clip(d, fc, f.p0); c1 = corner_clip(d, fc, f.p1); c2 = corner_clip(d, fc, f.p2);
    }
    const bool want_uv = kTextures && d.texture != RND_NONE;
    const float u0 = f.u0, v0 = f.v0, u1 = f.u1, v1 = f.v1, u2 = f.u2, v2 = f.v2;
    // perspective-correct barycentrics: 2D homogeneous edge functions of (x, y, w) at the pixel centre
    // (valid for w <= 0 vertices too)
#if RND_FAST_SHADE
    // f32: the triangle is first moved so the pixel centre is the origin (one FMA per coordinate,
    // so the differences of nearly equal numbers are exact up to the rounding of the centre itself,
    // ~1e-4 px), then the edge functions are plain 2x2 determinants of small numbers
    const float xn = fmaf((float)i + 0.5f, fc.ndc_step_xf, -1.0f);
    const float yn = fmaf(-((float)j + 0.5f), fc.ndc_step_yf, 1.0f);
    const float xr0 = fmaf(-xn, c0.w, c0.x), yr0 = fmaf(-yn, c0.w, c0.y);
    const float xr1 = fmaf(-xn, c1.w, c1.x), yr1 = fmaf(-yn, c1.w, c1.y);
    const float xr2 = fmaf(-xn, c2.w, c2.x), yr2 = fmaf(-yn, c2.w, c2.y);
    const float e0 = fmaf(xr1, yr2, -(xr2 * yr1));
    const float e1 = fmaf(xr2, yr0, -(xr0 * yr2));
    const float e2 = fmaf(xr0, yr1, -(xr1 * yr0));
    const float inv_s = rcp_f((e0 + e1) + e2);
    float b0 = e0 * inv_s, b1 = e1 * inv_s, b2 = e2 * inv_s;
    if (!(fabsf(b0) + fabsf(b1) + fabsf(b2) < 3.0e38f)) { b0 = 1.0f; b1 = 0.0f; b2 = 0.0f; }
#else
    double x0 = c0.x, y0 = c0.y, w0 = c0.w, x1 = c1.x, y1 = c1.y, w1 = c1.w, x2 = c2.x, y2 = c2.y, w2 = c2.w;
    double xn = ((double)i + 0.5) * fc.ndc_step_x - 1.0;
    double yn = 1.0 - ((double)j + 0.5) * fc.ndc_step_y;
    double e0 = ((y1 * w2 - w1 * y2) * xn + (w1 * x2 - x1 * w2) * yn) + (x1 * y2 - y1 * x2);
    double e1 = ((y2 * w0 - w2 * y0) * xn + (w2 * x0 - x2 * w0) * yn) + (x2 * y0 - y2 * x0);
    double e2 = ((y0 * w1 - w0 * y1) * xn + (w0 * x1 - x0 * w1) * yn) + (x0 * y1 - y0 * x1);
    double inv_s = 1.0 / ((e0 + e1) + e2);
    double d0 = e0 * inv_s, d1 = e1 * inv_s, d2 = e2 * inv_s;
    if (!(isfinite(d0) && isfinite(d1) && isfinite(d2))) { d0 = 1.0; d1 = 0.0; d2 = 0.0; }
    const float b0 = (float)d0, b1 = (float)d1, b2 = (float)d2;
#endif
    // attributes, interpolated in object space and transformed once
    float nx = (b0 * f.n0[0] + b1 * f.n1[0]) + b2 * f.n2[0];
    float ny = (b0 * f.n0[1] + b1 * f.n1[1]) + b2 * f.n2[1];
    float nz = (b0 * f.n0[2] + b1 * f.n1[2]) + b2 * f.n2[2];
    if (from_clip) {
        const float kx = (b0 * c0.x + b1 * c1.x) + b2 * c2.x, ky = (b0 * c0.y + b1 * c1.y) + b2 * c2.y,
                    kw = (b0 * c0.w + b1 * c1.w) + b2 * c2.w;
        const cptr_f32 q = fc.unproject;
        o.fx = ((q[0] * kx + q[3] * ky) + q[6] * kw) + fc.cam_pos[0];
        o.fy = ((q[1] * kx + q[4] * ky) + q[7] * kw) + fc.cam_pos[1];
        o.fz = ((q[2] * kx + q[5] * ky) + q[8] * kw) + fc.cam_pos[2];
    } else {
        const float px = (b0 * f.p0[0] + b1 * f.p1[0]) + b2 * f.p2[0];
        const float py = (b0 * f.p0[1] + b1 * f.p1[1]) + b2 * f.p2[1];
        const float pz = (b0 * f.p0[2] + b1 * f.p1[2]) + b2 * f.p2[2];
        const float* m = d.world;
        o.fx = ((m[0] * px + m[4] * py) + m[8] * pz) + m[12];
        o.fy = ((m[1] * px + m[5] * py) + m[9] * pz) + m[13];
        o.fz = ((m[2] * px + m[6] * py) + m[10] * pz) + m[14];
    }
    const float* q = d.nrm;
    float gx = (q[0] * nx + q[4] * ny) + q[8] * nz;
    float gy = (q[1] * nx + q[5] * ny) + q[9] * nz;
    float gz = (q[2] * nx + q[6] * ny) + q[10] * nz;
    // base colour = material.base_color * textureSample(...) (render_mesh.wgsl:89, :142); untextured
    // materials bind the 1x1 white texture (visual_server.rs:736)
    float tr = 1.0f, tg = 1.0f, tb = 1.0f;
    if (want_uv) {        // uniform over the wave (d is)
        const float uu = (b0 * u0 + b1 * u1) + b2 * u2, vv = (b0 * v0 + b1 * v1) + b2 * v2;
        // screen-space uv derivatives for the mip level, analytic: the edge functions are affine in
        // the pixel centre, b_k = e_k / S, so d b_k = (d e_k * S - e_k * dS) / S^2
        const float txn = fmaf((float)i + 0.5f, fc.ndc_step_xf, -1.0f), tyn = fmaf(-((float)j + 0.5f), fc.ndc_step_yf, 1.0f);
        const float px0 = fmaf(-txn, c0.w, c0.x), py0 = fmaf(-tyn, c0.w, c0.y);
        const float px1 = fmaf(-txn, c1.w, c1.x), py1 = fmaf(-tyn, c1.w, c1.y);
        const float px2 = fmaf(-txn, c2.w, c2.x), py2 = fmaf(-tyn, c2.w, c2.y);
        const float f0 = fmaf(px1, py2, -(px2 * py1)), f1 = fmaf(px2, py0, -(px0 * py2)), f2 = fmaf(px0, py1, -(px1 * py0));
        const float ax0 = c2.w * py1 - c1.w * py2, ax1 = c0.w * py2 - c2.w * py0, ax2 = c1.w * py0 - c0.w * py1;
        const float ay0 = px2 * c1.w - px1 * c2.w, ay1 = px0 * c2.w - px2 * c0.w, ay2 = px1 * c0.w - px0 * c1.w;
        const float S = (f0 + f1) + f2, Sx = (ax0 + ax1) + ax2, Sy = (ay0 + ay1) + ay2;
        const float is = 1.0f / S, is2 = is * is;
        const float kx = is2 * fc.ndc_step_xf, ky = is2 * (-fc.ndc_step_yf);
        float dx0 = (ax0 * S - f0 * Sx) * kx, dx1 = (ax1 * S - f1 * Sx) * kx, dx2 = (ax2 * S - f2 * Sx) * kx;
        float dy0 = (ay0 * S - f0 * Sy) * ky, dy1 = (ay1 * S - f1 * Sy) * ky, dy2 = (ay2 * S - f2 * Sy) * ky;
        if (!(fabsf(dx0) + fabsf(dx1) + fabsf(dx2) + fabsf(dy0) + fabsf(dy1) + fabsf(dy2) < 3.0e38f)) dx0 = dx1 = dx2 = dy0 = dy1 = dy2 = 0.0f;
        const float dudx = (dx0 * u0 + dx1 * u1) + dx2 * u2, dvdx = (dx0 * v0 + dx1 * v1) + dx2 * v2;
        const float dudy = (dy0 * u0 + dy1 * u1) + dy2 * u2, dvdy = (dy0 * v0 + dy1 * v1) + dy2 * v2;
        const f32x4 texel = texture_sample(textures[d.texture], lut, uu, vv, dudx, dvdx, dudy, dvdy);
        tr = texel.x; tg = texel.y; tb = texel.z;
    }
    o.br = d.base_color[0] * tr; o.bg = d.base_color[1] * tg; o.bb = d.base_color[2] * tb;
    o.lit = !(d.flags & kItemUnlit);
    if (o.lit) {
        float inv_n = rsqrt_f(dot3(gx, gy, gz, gx, gy, gz));
        o.Nx = gx * inv_n; o.Ny = gy * inv_n; o.Nz = gz * inv_n;
    } else {
        o.Nx = o.Ny = o.Nz = 0.0f;
    }
}

// what frag_finish leaves out, so that a finished pixel waits for its neighbour in ten registers, not sixteen: the
// ambient term and the unit vector to the camera
__device__ __forceinline__ void complete_inputs(PixelInputs& o, FrameConstRef fc) {
    if (o.lit) {
        float ar = fc.ambient[0] * fc.ambient[3], ag = fc.ambient[1] * fc.ambient[3], ab = fc.ambient[2] * fc.ambient[3];
        o.cr = o.br * ar; o.cg = o.bg * ag; o.cb = o.bb * ab;
        float vx = fc.cam_pos[0] - o.fx, vy = fc.cam_pos[1] - o.fy, vz = fc.cam_pos[2] - o.fz;
        float inv_v = rsqrt_f(dot3(vx, vy, vz, vx, vy, vz));
        o.Vx = vx * inv_v; o.Vy = vy * inv_v; o.Vz = vz * inv_v;
    } else {
        o.cr = o.br; o.cg = o.bg; o.cb = o.bb;
        o.Vx = o.Vy = o.Vz = 0.0f;
    }
}

// the three stages in a row, for one pixel
template <bool kTextures, bool kStored>
__device__ __forceinline__ void prepare_pixel(const DrawItem& d, FrameConstRef fc, const float4* __restrict__ clip,
                                              const TextureDesc* __restrict__ textures, const float* lut, uint32_t id, uint32_t i,
                                              uint32_t j, PixelInputs& o, FrameCounters* counters) {
    FragFetch f;
    bool ok;
    frag_indices(d, id, f, ok);
    frag_fetch<kTextures, kStored>(d, fc, clip, f, ok, counters);
    frag_finish<kTextures, kStored>(d, fc, clip, textures, lut, i, j, f, o);
}

// compute_light_occlusion (render_mesh.wgsl:184-252): cascade by view depth, biased position
// through the cascade's matrix, 3x3 nearest taps (ClampToBorder, OpaqueWhite: depth 1 outside the
// map, backend.rs:441-453).  The shader can select layer 3 although the uniform holds three
// matrices (:118, :203): with naga's index clamping that layer is read through matrix 2.
__device__ __forceinline__ float light_occlusion(const ShadowSet* __restrict__ sets, int32_t set, FrameConstRef fc,
                                                 float fx, float fy, float fz, float nx, float ny, float nz,
                                                 float lx, float ly, float lz) {
    if (set < 0) return 0.0f;
    if (dot3(nx, ny, nz, lx, ly, lz) > 0.0f) return 0.0f;
    const ShadowSet& S = sets[set];
    const float view_depth = ((fc.view[2] * fx + fc.view[6] * fy) + fc.view[10] * fz) + fc.view[14];
    uint32_t layer; float bias_mod;
    if (view_depth <= 0.05f * 100.0f) { layer = 0; bias_mod = 0.03f; }
    else if (view_depth <= 0.1f * 100.0f) { layer = 1; bias_mod = 0.05f; }
    else if (view_depth <= 0.3f * 100.0f) { layer = 2; bias_mod = 0.1f; }
    else { layer = 3; bias_mod = 0.2f; }
    const float ox = ((-lx) * 0.3f + nx * 0.8f) * bias_mod, oy = ((-ly) * 0.3f + ny * 0.8f) * bias_mod,
                oz = ((-lz) * 0.3f + nz * 0.8f) * bias_mod;
    const float bx = fx + ox, by = fy + oy, bz = fz + oz;
    const float* m = S.world_to_light[layer < 2u ? layer : 2u];
    const float cx = ((m[0] * bx + m[4] * by) + m[8] * bz) + m[12] * 1.0f;
    const float cy = ((m[1] * bx + m[5] * by) + m[9] * bz) + m[13] * 1.0f;
    const float cz = ((m[2] * bx + m[6] * by) + m[10] * bz) + m[14] * 1.0f;
    const float cw = ((m[3] * bx + m[7] * by) + m[11] * bz) + m[15] * 1.0f;
    const float qx = cx / cw, qy = cy / cw, qz = cz / cw;
    const float u = (qx + 1.0f) * 0.5f, v = 1.0f - (qy + 1.0f) * 0.5f;
    const float frag_depth = fminf(fmaxf(qz, 0.0f), 1.0f);
    const float texel = 1.0f / (float)kShadowSize;
    const uint64_t* __restrict__ map = S.layers[layer];
    constexpr uint32_t kTilesX = kShadowSize / kTile;
    float occlusion = 0.0f;
#pragma unroll
    for (int x = -1; x <= 1; ++x)
#pragma unroll
        for (int y = -1; y <= 1; ++y) {
            const float su = u + (float)x * texel, sv = v + (float)y * texel;
            float occluder = 1.0f;
            if (su >= 0.0f && su < 1.0f && sv >= 0.0f && sv < 1.0f) {
                const uint32_t ix = min((uint32_t)floorf(su * (float)kShadowSize), kShadowSize - 1u);
                const uint32_t iy = min((uint32_t)floorf(sv * (float)kShadowSize), kShadowSize - 1u);
                const size_t at = (size_t)((iy / kTile) * kTilesX + ix / kTile) * kTilePixels + ((iy % kTile) * kTile + ix % kTile);
                occluder = __uint_as_float((uint32_t)(map[at] >> 32));
            }
            if (frag_depth > occluder) occlusion += 1.0f;
        }
    return occlusion / 9.0f;
}

// The same look-up for the two fragments of a lane, straight-line: both fragments' nine taps are in flight together
// (18 four-byte loads of the depth halves), the cascade's matrix and layer pointer come from the LDS copy of the
// set, and the early exits are predicates (a lane without a lit fragment, or facing away, computes on whatever it
// holds and its result is dropped; every address is clamped into the map).  Same arithmetic per fragment as
// light_occlusion, in the same order.
constexpr uint32_t kLdsShadowSets = 8;
struct ShadowProbe { float u, v, frag_depth; const uint32_t* map; bool facing; };
__device__ __forceinline__ ShadowProbe shadow_probe(const ShadowSet& S, FrameConstRef fc, float fx, float fy, float fz, float nx, float ny,
                                                    float nz, float lx, float ly, float lz) {
    ShadowProbe q;
    q.facing = !(dot3(nx, ny, nz, lx, ly, lz) > 0.0f);
    const float view_depth = ((fc.view[2] * fx + fc.view[6] * fy) + fc.view[10] * fz) + fc.view[14];
    uint32_t layer; float bias_mod;
    if (view_depth <= 0.05f * 100.0f) { layer = 0; bias_mod = 0.03f; }
    else if (view_depth <= 0.1f * 100.0f) { layer = 1; bias_mod = 0.05f; }
    else if (view_depth <= 0.3f * 100.0f) { layer = 2; bias_mod = 0.1f; }
    else { layer = 3; bias_mod = 0.2f; }
    const float ox = ((-lx) * 0.3f + nx * 0.8f) * bias_mod, oy = ((-ly) * 0.3f + ny * 0.8f) * bias_mod,
                oz = ((-lz) * 0.3f + nz * 0.8f) * bias_mod;
    const float bx = fx + ox, by = fy + oy, bz = fz + oz;
    const float* m = S.world_to_light[layer < 2u ? layer : 2u];
    const float cx = ((m[0] * bx + m[4] * by) + m[8] * bz) + m[12] * 1.0f;
    const float cy = ((m[1] * bx + m[5] * by) + m[9] * bz) + m[13] * 1.0f;
    const float cz = ((m[2] * bx + m[6] * by) + m[10] * bz) + m[14] * 1.0f;
    const float cw = ((m[3] * bx + m[7] * by) + m[11] * bz) + m[15] * 1.0f;
    const float qx = cx / cw, qy = cy / cw, qz = cz / cw;
    q.u = (qx + 1.0f) * 0.5f; q.v = 1.0f - (qy + 1.0f) * 0.5f;
    q.frag_depth = fminf(fmaxf(qz, 0.0f), 1.0f);
    q.map = reinterpret_cast<const uint32_t*>(S.layers[layer]);
    return q;
}
__device__ __forceinline__ void shadow_taps_issue(const ShadowProbe& q, uint32_t (&tap)[9], uint32_t& inside) {
    const float texel = 1.0f / (float)kShadowSize;
    constexpr uint32_t kTilesX = kShadowSize / kTile;
    inside = 0;
#pragma unroll
    for (int x = -1; x <= 1; ++x)
#pragma unroll
        for (int y = -1; y <= 1; ++y) {
            const int k = (x + 1) * 3 + (y + 1);
            const float su = q.u + (float)x * texel, sv = q.v + (float)y * texel;
            if (su >= 0.0f && su < 1.0f && sv >= 0.0f && sv < 1.0f) inside |= 1u << k;
            const uint32_t ix = min((uint32_t)fmaxf(floorf(su * (float)kShadowSize), 0.0f), kShadowSize - 1u);
            const uint32_t iy = min((uint32_t)fmaxf(floorf(sv * (float)kShadowSize), 0.0f), kShadowSize - 1u);
            const uint32_t at = ((iy / kTile) * kTilesX + ix / kTile) * kTilePixels + ((iy % kTile) * kTile + ix % kTile);
            tap[k] = q.map[2u * at + 1u];       // the depth half of the visibility word
        }
}
__device__ __forceinline__ float shadow_taps_count(const ShadowProbe& q, const uint32_t (&tap)[9], uint32_t inside) {
    float occlusion = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {           // (same order as the loops of light_occlusion: x outer, y inner)
        const float occluder = ((inside >> k) & 1u) ? __uint_as_float(tap[k]) : 1.0f;
        if (q.frag_depth > occluder) occlusion += 1.0f;
    }
    return q.facing ? occlusion / 9.0f : 0.0f;
}

// fs_main_blinn_phong for one light and one fragment (render_mesh.wgsl:134-181, :256-281)
template <bool kShadows>
__device__ __forceinline__ void add_light(const GpuLight& L, PixelInputs& p, const ShadowSet* __restrict__ sets,
                                          FrameConstRef fc) {
    float dx, dy, dz, intensity;
    if (L.kind == RND_LIGHT_DIRECTIONAL) {
        dx = L.dir[0]; dy = L.dir[1]; dz = L.dir[2];
        float occlusion = 0.0f;
        if constexpr (kShadows) occlusion = light_occlusion(sets, L.shadow_index, fc, p.fx, p.fy, p.fz, p.Nx, p.Ny, p.Nz, dx, dy, dz);
        intensity = L.color[3] * (1.0f - occlusion);
    } else {
        float ox = p.fx - L.pos[0], oy = p.fy - L.pos[1], oz = p.fz - L.pos[2];
        float d2 = dot3(ox, oy, oz, ox, oy, oz);
#if RND_FAST_SHADE
        float inv_d = rsqrt_f(d2);
        float dist = d2 * inv_d;
        if (!(dist <= L.radius)) return;
        float lin = fminf(fmaxf((L.radius - dist) * L.inv_radius, 0.0f), 1.0f);
#else
        float dist = sqrtf(d2);
        if (dist > L.radius) return;
        float inv_d = 1.0f / dist;
        float lin = fminf(fmaxf((L.radius - dist) / L.radius, 0.0f), 1.0f);
#endif
        dx = ox * inv_d; dy = oy * inv_d; dz = oz * inv_d;
        float att = (lin * lin) * (3.0f - 2.0f * lin);
        intensity = L.color[3] * att;
    }
    float tx = -dx, ty = -dy, tz = -dz;
    float diff = fmaxf(dot3(p.Nx, p.Ny, p.Nz, tx, ty, tz), 0.0f);
    float lr = L.color[0] * intensity, lg = L.color[1] * intensity, lb = L.color[2] * intensity;
    float hx = tx + p.Vx, hy = ty + p.Vy, hz = tz + p.Vz;
    float inv_h = rsqrt_f(dot3(hx, hy, hz, hx, hy, hz));
    hx *= inv_h; hy *= inv_h; hz *= inv_h;
    float sp = fmaxf(dot3(p.Nx, p.Ny, p.Nz, hx, hy, hz), 0.0f);
    float s2 = sp * sp, s4 = s2 * s2, s8 = s4 * s4;
    p.cr = p.cr + p.br * (lr * diff + lr * s8);
    p.cg = p.cg + p.bg * (lg * diff + lg * s8);
    p.cb = p.cb + p.bb * (lb * diff + lb * s8);
}

#if RND_FAST_SHADE
// Two fragments per lane: the same Blinn-Phong terms as add_light with products and sums contracted to FMAs and
// the colour factors regrouped; colour parity is a tolerance.
struct PixelPair {
    f32x2 fx, fy, fz, Nx, Ny, Nz, Vx, Vy, Vz, br, bg, bb, cr, cg, cb;
    f32x2 NV;               // N . V
    f32x2 ar, ag, ab;       // sum over the lights of colour * intensity * (diffuse + specular)
    bool lit0, lit1;
};
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ f32x2 rsq2(f32x2 v) { return (f32x2){__builtin_amdgcn_rsqf(v.x), __builtin_amdgcn_rsqf(v.y)}; }
__device__ __forceinline__ f32x2 max0(f32x2 v) { return (f32x2){fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f)}; }
__device__ __forceinline__ f32x2 sat2(f32x2 v) { return (f32x2){fminf(fmaxf(v.x, 0.0f), 1.0f), fminf(fmaxf(v.y, 0.0f), 1.0f)}; }

// A light as the shading kernel keeps it in LDS (32 bytes, two 16-byte reads): position and radius of a point light,
// or the (unnormalized) direction of a directional one with a negative radius as its mark; the colour already times
// the intensity.  The light loop reads the current light with all lanes at one address (a broadcast): its fields
// arrive in VGPRs, and the loop's arithmetic has no SGPR operand — on gfx950 an f32 FMA / multiply / add with an SGPR
// operand issues at half the rate of one on VGPRs alone (profiles/r02_valu_ops_microbench.txt), and the 32 SGPRs that
// the scalar-load version held for the current and the next light made the kernel spill SGPRs into VGPR lanes.
struct ShadeLight { float x, y, z, radius, r, g, b, inv_radius; };
static_assert(sizeof(ShadeLight) == 32, "ShadeLight is 2 x 16 bytes");

// The lane's lit flags are not looked at here (except to skip shadow look-ups): a lane without a lit fragment
// computes on whatever it holds and the caller drops its sum.  Straight-line for a point light: beyond the radius the
// clamped linear term is zero and so is the light's weight, which is what the reference's discard amounts to.
//   o = frag - light (unnormalized), dist = |o|:   diffuse = max(-(N.o) / dist, 0)
//   half vector scaled by dist: h' = V * dist - o;  specular base = max((N.h') / |h'|, 0), with N.h' = (N.V) dist - N.o
// Cosines of unit vectors are clamped to [0, 1] instead of max(., 0): the same value up to the last bit of a cosine
// that rounds above one, and the clamp rides on the multiply that produces it.  A point light's diffuse term is such
// a cosine; a directional light's is not (its direction is not normalized, as in the shader), so that one keeps the
// maximum.  `inten`: attenuation (point) or 1 - occlusion (directional); the light's colour carries its intensity.
template <bool kPoint>
__device__ __forceinline__ void light_pair_finish(PixelPair& p, float cr, float cg, float cb, f32x2 ox, f32x2 oy, f32x2 oz, f32x2 inv_d,
                                                  f32x2 dist, f32x2 inten) {
#pragma clang fp contract(fast)
    const f32x2 No = p.Nx * ox + p.Ny * oy + p.Nz * oz;
    f32x2 diff;
    if constexpr (kPoint) diff = sat2(-(No * inv_d));
    else diff = max0(-No);
    const f32x2 hx = p.Vx * dist - ox, hy = p.Vy * dist - oy, hz = p.Vz * dist - oz;
    const f32x2 inv_h = rsq2(hx * hx + hy * hy + hz * hz);
    const f32x2 sp = sat2((p.NV * dist - No) * inv_h);
    const f32x2 s2 = sp * sp, s4 = s2 * s2;
    // the lights' sum is kept apart from the base colour and multiplied in once after the loop
    const f32x2 w = (s4 * s4 + diff) * inten;
    p.ar += splat2(cr) * w;
    p.ag += splat2(cg) * w;
    p.ab += splat2(cb) * w;
}
__device__ __forceinline__ void add_point_pair(const f32x4 a, const f32x4 b, PixelPair& p) {
#pragma clang fp contract(fast)
    const f32x2 ox = p.fx - splat2(a.x), oy = p.fy - splat2(a.y), oz = p.fz - splat2(a.z);
    const f32x2 d2 = ox * ox + oy * oy + oz * oz;
    const f32x2 inv_d = rsq2(d2);
    const f32x2 dist = d2 * inv_d;
    const f32x2 lin = sat2(splat2(1.0f) - dist * splat2(b.w));
    light_pair_finish<true>(p, b.x, b.y, b.z, ox, oy, oz, inv_d, dist, (lin * lin) * (splat2(3.0f) - splat2(2.0f) * lin));
}
// a directional light: a = its direction, b.xyz = colour * intensity, `set` = its cascade set or -1
template <bool kShadows>
__device__ __forceinline__ void add_directional_pair(const f32x4 a, const f32x4 b, int32_t set, PixelPair& p,
                                                     const ShadowSet* __restrict__ sets, const ShadowSet* lds_sets, FrameConstRef fc) {
    float i0 = 1.0f, i1 = 1.0f;
    if constexpr (kShadows) {
        if (set >= 0) {                                            // (uniform over the wave)
            if ((uint32_t)set >= kLdsShadowSets) {                 // beyond the LDS copy: one by one
                if (p.lit0) i0 = 1.0f - light_occlusion(sets, set, fc, p.fx.x, p.fy.x, p.fz.x, p.Nx.x, p.Ny.x, p.Nz.x, a.x, a.y, a.z);
                if (p.lit1) i1 = 1.0f - light_occlusion(sets, set, fc, p.fx.y, p.fy.y, p.fz.y, p.Nx.y, p.Ny.y, p.Nz.y, a.x, a.y, a.z);
            } else {
                const ShadowSet& S = lds_sets[set];
                const ShadowProbe q0 = shadow_probe(S, fc, p.fx.x, p.fy.x, p.fz.x, p.Nx.x, p.Ny.x, p.Nz.x, a.x, a.y, a.z);
                const ShadowProbe q1 = shadow_probe(S, fc, p.fx.y, p.fy.y, p.fz.y, p.Nx.y, p.Ny.y, p.Nz.y, a.x, a.y, a.z);
                uint32_t t0[9], t1[9], in0, in1;
                shadow_taps_issue(q0, t0, in0);
                shadow_taps_issue(q1, t1, in1);
                i0 = 1.0f - shadow_taps_count(q0, t0, in0);
                i1 = 1.0f - shadow_taps_count(q1, t1, in1);
            }
        }
    }
    light_pair_finish<false>(p, b.x, b.y, b.z, splat2(a.x), splat2(a.y), splat2(a.z), splat2(1.0f), splat2(1.0f), (f32x2){i0, i1});
}
#endif

// one light through scalar loads (its index is wave-uniform): the per-fragment path of the builds without RND_FAST_SHADE
__device__ __forceinline__ GpuLight load_light_uniform(const GpuLight* lights, uint32_t li) {
    typedef const __attribute__((address_space(4))) GpuLight* cptr_light;
    cptr_light lp = (cptr_light)lights + li;
    GpuLight L;
#pragma unroll
    for (int c = 0; c < 3; ++c) { L.pos[c] = lp->pos[c]; L.dir[c] = lp->dir[c]; }
#pragma unroll
    for (int c = 0; c < 4; ++c) L.color[c] = lp->color[c];
    L.radius = lp->radius; L.kind = lp->kind; L.shadow_index = lp->shadow_index; L.inv_radius = lp->inv_radius;
    L.pad[0] = L.pad[1] = 0;
    return L;
}

// -DRND_SHADE_PROFILE: per-phase cycle stamps of k_shade's waves, summed into g_shade_prof (64 slots x 16 phases);
// shade_profile_dump() prints cycles per wave and phase (called when a context is destroyed).  Development aid only.
#ifdef RND_SHADE_PROFILE
__device__ unsigned long long g_shade_prof[64][16];
#define RND_PROF_DECL unsigned long long prof_t_ = clock64(); unsigned long long prof_acc_[12] = {}
#define RND_PROF(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = clock64(); prof_acc_[k] += t_ - prof_t_; prof_t_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define RND_PROF_FLUSH do { if (lane == 0) { for (int q_ = 0; q_ < 12; ++q_) atomicAdd(&g_shade_prof[blockIdx.x & 63u][q_], prof_acc_[q_]); atomicAdd(&g_shade_prof[blockIdx.x & 63u][15], 1ull); } } while (0)
void shade_profile_dump() {
    static unsigned long long h[64][16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_shade_prof), sizeof h) != hipSuccess) return;
    unsigned long long sum[16] = {};
    for (auto& row : h) for (int k = 0; k < 16; ++k) sum[k] += row[k];
    if (!sum[15]) return;
    fprintf(stderr, "k_shade profile: %llu waves; cycles per wave by phase:", sum[15]);
    unsigned long long all = 0;
    for (int k = 0; k < 12; ++k) { fprintf(stderr, " [%d] %.0f", k, (double)sum[k] / (double)sum[15]); all += sum[k]; }
    fprintf(stderr, "  total %.0f\n", (double)all / (double)sum[15]);
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_shade_prof), h, sizeof h);
}
#else
#define RND_PROF_DECL
#define RND_PROF(k)
#define RND_PROF_FLUSH
void shade_profile_dump() {}
#endif

// A workgroup shades one part of a tile: 512 consecutive pixels (16 rows), two pixels per lane.
// kShadows ("full" variant): the frame has directional lights with cascade sets or textured draw
// items; both cost registers, so frames without them run the variant compiled without them.
// kSamples: 1, or 4 (MSAA; a part is 256 pixels): phases A-C run once per pair of sample planes — the fragment of the
// sample's winning triangle is shaded at the pixel centre, exactly as the reference's per-pixel fragment stage,
// and re-used while consecutive samples hold the same triangle — each sample keeps the f16 colour,
// and the pixel gets their mean (the resolve attachment, visual_server.rs:1134-1153).
// Dynamic LDS: the frame's lights as ShadeLight (32 B each), then whatever the launch adds to set the occupancy.
template <bool kShadows, int kSamples, bool kStored>
// (waves per SIMD the variants are compiled for: the plain one 7, the full one 5 — 95 VGPRs without a spill where the
// compiler left alone took 101 and with them a fifth wave: C3 + sun shade 0.268 -> 0.241 ms —, plain MSAA 5 (seven
// spilled dwords at 96 VGPRs against 102 at four waves: C3 MSAA x4 shade 0.72 -> 0.68 ms); the full MSAA variant spills
// 29 at 4 and keeps its 3)
__global__ __launch_bounds__(256, kShadows ? (kSamples > 1 ? 1 : 5) : (kSamples > 1 ? 5 : 7)) void k_shade(FrameBuffers fb, uint32_t n_parts) {
    FrameConstRef fc = frame_const_ref(fb.frame_const);
    constexpr uint32_t PPT = 2;                                         // fragments per lane in flight: two pixels, or two samples of one
    constexpr uint32_t kPix = kSamples > 1 ? 1u : PPT;                  // pixels per lane
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ uint32_t s_bases[kLdsItems + 1];
    __shared__ float s_lut[kShadows ? 512 : 1];     // texel decode table (see FetchSrgb8), full variant only
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // The frame's shape comes by value (FrameBuffers::dims), not from the constants in device memory: what a workgroup
    // needs before its first loads can go out — the visibility words, the tables for LDS, the counters — then depends
    // on the kernel arguments alone and is one round trip, where the tile list and the constants were two more in
    // front of it (a workgroup's set-up was 11 k of the 33 k cycles a wave spends on a part).
    const uint32_t n_lights = min(fb.dims.n_lights, kMaxLights), n_items = fb.dims.n_items;
    ShadeLight* const s_light = reinterpret_cast<ShadeLight*>(s_dyn);
    // Part p = blockIdx.x: XCD p % 8 (where the workgroup runs),
    // then which sixteen rows of the tile, then the tile's number on its XCD: every XCD shades one contiguous run of
    // the row-major list of owned tiles (rebuild_tiles keeps the same order for the other kernels), in closed form.
    constexpr uint32_t kPartShift = kSamples > 1 ? 2u : 1u;            // log2(parts per tile)
    auto locate = [&](uint32_t p, uint32_t& tile, uint32_t& compact, uint32_t& first) {
        const uint32_t xcd = p & (kXcds - 1u), sub = (p >> 3) & ((1u << kPartShift) - 1u), j = p >> (3u + kPartShift);
        const uint32_t q = fb.n_owned >> 3, r = fb.n_owned & (kXcds - 1u);
        const uint32_t idx = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + j;
        compact = idx; first = sub * (256u * kPix);
        tile = idx;
        if (fb.dims.world > 1u) {
            // the idx-th owned tile of a stripe partition: full-width stripes of kOwnerGroup tile rows, dealt round-robin
            const uint32_t row = fb.dims.tiles_x_magic ? __umulhi(idx, fb.dims.tiles_x_magic) : idx, tx = idx - row * fb.dims.tiles_x;
            // (rank r holds slot world - 1 - r of every round: stripe_owner)
            tile = (((row / kOwnerGroup) * fb.dims.world + (fb.dims.world - 1u - fb.dims.rank)) * kOwnerGroup + row % kOwnerGroup) * fb.dims.tiles_x + tx;
        }
        return p < n_parts && j < q + (xcd < r ? 1u : 0u);       // (the last tiles of an XCD's run may not exist)
    };
    auto part_vis = [&](uint32_t tile, uint32_t first, uint32_t smp_plane, uint32_t k) {
        const unsigned long long* v = reinterpret_cast<const unsigned long long*>(fb.vis) + (size_t)tile * kTilePixels + (size_t)smp_plane * fb.vis_stride;
        return v[first + wave * (64u * kPix) + (kSamples > 1 ? 0u : k * 64u) + lane];
    };
    RND_PROF_DECL;
    uint32_t tile_s, compact_s, part;
    if (!locate(blockIdx.x, tile_s, compact_s, part)) return;          // (uniform over the workgroup)
    // The visibility words of the lane's pixels (of its first two samples under MSAA) are requested before anything else:
    // the first of the fragment phase's dependent round trips then runs beside the workgroup's set-up (tables to LDS,
    // counters, barrier) instead of after it.
    unsigned long long key_first[PPT];
#pragma unroll
    for (uint32_t k = 0; k < PPT; ++k) key_first[k] = part_vis(tile_s, part, kSamples > 1 ? k : 0u, k);
    // The tables for LDS: every load first, then one wait, then the LDS writes and the barrier — ONE round trip.  (The
    // copy loops of round 2 each waited for their own loads, and with them for the visibility words above: four round
    // trips in a row under the kernel's own gather traffic, 11 k of the 33 k cycles of a wave's life.)
    const bool bases_in_lds = n_items <= kLdsItems;
    uint32_t base_v = 0u;
    if (bases_in_lds && tid <= n_items) base_v = fb.tri_bases[tid];
    f32x4 l0 = {0.0f, 0.0f, 0.0f, 0.0f}, l1 = l0, l2 = l0, l3 = l0;
    if (tid < n_lights) {
        const f32x4* lp = reinterpret_cast<const f32x4*>(fb.lights + tid);
        l0 = lp[0]; l1 = lp[1]; l2 = lp[2]; l3 = lp[3];
    }
    float lut0 = 0.0f, lut1 = 0.0f;
    if constexpr (kShadows) { lut0 = fb.srgb_lut[tid]; lut1 = fb.srgb_lut[256u + tid]; }
    __shared__ ShadowSet s_sets[kShadows ? kLdsShadowSets : 1];      // the first cascade sets (matrices, layer pointers), full variant only
    // A frame whose geometry phase overflowed an internal stream (records, clip queue, bin entries: triangles were
    // dropped) is not presented: nothing is written, the image keeps the previous frame.  The host learns of it
    // from the report (no synchronization) and has grown the streams by the next frame; rnd_finish re-runs it.
    {
        typedef const __attribute__((address_space(4))) FrameCounters* cptr_counters;      // scalar loads
        uint32_t ovf = 0, n_rec = 0, n_clip = 0, n_ent = 0, need = 0;
        for (uint32_t s = 0; s < fb.n_counter_sets; ++s) {
            if (s < 64u && !((fb.ran_mask >> s) & 1ull)) continue;
            cptr_counters c = (cptr_counters)fb.all_counters + s;
            ovf |= c->overflow; n_rec = max(n_rec, c->n_records); n_clip = max(n_clip, c->n_clip); n_ent = max(n_ent, c->n_entries);
            need |= c->chain_needed;
        }
        // A frame launched without the record chain (the host does that after a run of frames that needed none of it) whose
        // setup did write records or queue triangles for the clipper is withheld like an overflowed one, and re-run in full.
        const bool no_chain_failed = fb.no_chain != 0u && need != 0u;
        if (blockIdx.x == 0 && fb.report) {
            uint32_t frags = ((cptr_counters)fb.main_counters)->n_fragments;       // (the main pass's: the host picks its k_setup variant by it)
            if (fb.no_chain) {
                // k_scan_tiles did not run: the main pass's statistics are summed here, one stripe per lane of the first
                // wave (vector loads on purpose: the compiler hoists scalar loads of constant-address-space pointers
                // above the branch, and then every workgroup of the grid pays for 32 cache lines: +17 us at C3)
                __shared__ unsigned long long s_sum[2];
                if (tid < 64u) {
                    const unsigned long long w = tid < kStatStripes ? __builtin_nontemporal_load(&fb.stat_stripes[tid * kStatStride]) : 0ull;
                    unsigned long long v = w & 0xFFFFFFFFull, fr = w >> 32;
                    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); fr += __shfl_xor(fr, o); }
                    if (tid == 0) { s_sum[0] = v; s_sum[1] = fr; }
                }
                __syncthreads();
                const unsigned long long v = s_sum[0], fr = s_sum[1];
                frags = (uint32_t)min(fr, 0xFFFFFFFFull);
                if (tid == 0) { fb.counters->n_valid = (uint32_t)min(v, 0xFFFFFFFFull); fb.counters->n_fragments = frags; }
            }
            if (tid == 0) {
                volatile FrameReport* r = fb.report;
                r->overflow = (ovf & 7u) | (no_chain_failed ? kReportNoChainFailed : 0u) | (need ? kReportChainNeeded : 0u);
                r->n_records = n_rec; r->n_clip = n_clip; r->n_entries = n_ent;
                r->n_fragments = frags;
                __threadfence_system();
                r->frame_tag = fb.report_tag;
            }
        }
        if ((ovf & 7u) || no_chain_failed) return;          // (uniform over the grid; before any barrier the whole grid takes)
    }
    if constexpr (kShadows) {
        s_lut[tid] = lut0; s_lut[256u + tid] = lut1;
        const uint32_t n_words = min(fb.n_shadow_sets, kLdsShadowSets) * (uint32_t)(sizeof(ShadowSet) / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(fb.shadow_sets);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_sets);
        for (uint32_t k = tid; k < n_words; k += 256u) dst[k] = src[k];
    }
    if (bases_in_lds) {
        if (tid <= n_items) s_bases[tid] = base_v;
        for (uint32_t k = tid + 256u; k <= n_items; k += 256u) s_bases[k] = fb.tri_bases[k];
    }
    static_assert(kMaxLights <= 256, "one thread per light");
    static_assert(sizeof(GpuLight) == 64, "a light is four 16-byte loads");
    if (tid < n_lights) {
        // l0 = pos.xyz, radius; l1 = dir.xyz, kind; l2 = colour rgb, intensity; l3 = shadow_index, inv_radius, -, -
        const bool directional = __float_as_uint(l1.w) == RND_LIGHT_DIRECTIONAL;
        f32x4* dst = reinterpret_cast<f32x4*>(s_light + tid);
        dst[0] = directional ? (f32x4){l1.x, l1.y, l1.z, -1.0f} : (f32x4){l0.x, l0.y, l0.z, fmaxf(l0.w, 0.0f)};
        dst[1] = (f32x4){l2.x * l2.w, l2.y * l2.w, l2.z * l2.w, l3.y};
    }
    __syncthreads();
    RND_PROF(0);

  {
    const uint32_t tile_ty = fb.dims.tiles_x_magic ? __umulhi(tile_s, fb.dims.tiles_x_magic) : tile_s;
    const uint32_t tile_x0 = (tile_s - tile_ty * fb.dims.tiles_x) * kTile, tile_y0 = tile_ty * kTile;
    const unsigned long long* vis0 = reinterpret_cast<const unsigned long long*>(fb.vis) + (size_t)tile_s * kTilePixels;
    PixelInputs px[PPT];
    unsigned long long key[PPT];
    bool covered[PPT];
    float sum_r = 0.0f, sum_g = 0.0f, sum_b = 0.0f;     // MSAA: running sum of the samples' f16 colours
    uint32_t prev_id[PPT];
    bool reuse[PPT];
#pragma unroll
    for (uint32_t k = 0; k < PPT; ++k) { prev_id[k] = RND_NONE; reuse[k] = false; }
    float keep[PPT][3] = {};        // MSAA: the colour each of the lane's two samples was shaded to in the previous pair
    unsigned long long key0 = 0;
  // MSAA: the lane's two "pixels" are two samples of ONE pixel (samples 0, 1, then 2, 3), so that the light loop runs
  // on the pair path as it does for two pixels
  for (int smp = 0; smp < kSamples; smp += (kSamples > 1 ? PPT : 1)) {
    // phase A: fragment inputs of this lane's pixels.  The visibility words of all of them first, then — the usual
    // case: every covered pixel of the wave, over all of the lane's pixels, belongs to ONE draw item — each stage of
    // the fragment fetch for all pixels before the next stage (the loaded data of two pixels is 42
    // registers), so that the lane's dependent round trips are four (word, item, indices, corners) and not four per pixel.
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t id[PPT], pi[PPT], pj[PPT], it[PPT];
    PixelInputs fresh[PPT] = {};
#pragma unroll
    for (uint32_t k = 0; k < PPT; ++k) {
        const unsigned long long* vis = vis0 + (size_t)(kSamples > 1 ? (uint32_t)smp + k : 0u) * fb.vis_stride;
        uint32_t local = part + wave * (64u * kPix) + (kSamples > 1 ? 0u : k * 64u) + lane;      // a wave holds 2*kPix consecutive rows
        pi[k] = tile_x0 + (local & (kTile - 1)); pj[k] = tile_y0 + (local / kTile);
        key[k] = smp == 0 ? key_first[k] : vis[local];
    }
#pragma unroll
    for (uint32_t k = 0; k < PPT; ++k) {
        id[k] = (uint32_t)(key[k] & 0xFFFFFFFFull);
        if constexpr (kSamples > 1) {
            if (smp == 0 && k == 0) key0 = key[k];
            // same triangle as two samples ago: same fragment (shaded at the pixel centre), same colour: px[k] still holds it
            reuse[k] = smp > 0 && id[k] == prev_id[k];
            prev_id[k] = id[k];
        }
        covered[k] = pi[k] < fb.dims.width && pj[k] < fb.dims.height && id[k] != RND_NONE;
        if constexpr (kSamples > 1) covered[k] = covered[k] && !reuse[k];
        if (covered[k] && !guard(id[k] < fc.n_tris, fb.counters, kGuardShade)) covered[k] = false;
        // the fragment's inputs go to `fresh`, left unset on lanes without a fragment; the defaults are
        // selected once, after the item paths have merged (initializing up front made every merge point
        // of the branches below re-materialize them)
        it[k] = 0;
    }
    bool staged = false;
    {
        const bool any_k = covered[0] || covered[1];
        const unsigned long long cov = __ballot(any_k);
        RND_PROF(1);
        if (cov) {
            // the draw item of the wave's first fragment, looked up once for the wave (all lanes compare one range of
            // the base table each; tables of more than 64 items: that lane's own binary search), then every other
            // fragment is tested against its triangle range: two instructions a pixel instead of a search
            uint32_t lead_s;
            if (bases_in_lds && n_items <= 64u) {
                uint32_t id_s;
                uniform_pick(covered[0] ? id[0] : id[1], cov, id_s);
                lead_s = find_item_wave(s_bases, n_items, id_s, lane);
                asm volatile("" : "+s"(lead_s));
            } else {
                const uint32_t some = covered[0] ? id[0] : id[1];
                uniform_pick(any_k ? (bases_in_lds ? find_item(s_bases, n_items, some) : find_item(fb.tri_bases, n_items, some)) : 0u, cov, lead_s);
            }
            if (lead_s < n_items) {
                const DrawItem d = load_item_uniform(fb.items, lead_s);
                const bool others = (covered[0] && id[0] - d.tri_base >= d.n_tris) || (covered[1] && id[1] - d.tri_base >= d.n_tris);
                if (__ballot(others) == 0ull && world_from_clip(d, fc)) {
                    staged = true;
                    FragFetch f[PPT];
                    bool ok[PPT];
                    // (a pixel without a fragment fetches the item's first triangle and drops it)
#pragma unroll
                    for (uint32_t k = 0; k < PPT; ++k) frag_indices(d, covered[k] ? id[k] : d.tri_base, f[k], ok[k]);
                    RND_PROF(2);
#pragma unroll
                    for (uint32_t k = 0; k < PPT; ++k) frag_fetch<kShadows, kStored>(d, fc, fb.clip, f[k], ok[k] || !covered[k], fb.counters);
                    RND_PROF(3);
                    // (for every lane: one without a fragment finishes the triangle it fetched, straight-line)
#pragma unroll
                    for (uint32_t k = 0; k < PPT; ++k) frag_finish<kShadows, kStored>(d, fc, fb.clip, fb.textures, s_lut, pi[k], pj[k], f[k], fresh[k]);
                    RND_PROF(4);
                }
            }
        } else staged = true;        // nothing covered
    }
    if (!staged) {
#pragma unroll
        for (uint32_t k = 0; k < PPT; ++k)
            if (covered[k]) it[k] = bases_in_lds ? find_item(s_bases, n_items, id[k]) : find_item(fb.tri_bases, n_items, id[k]);
#pragma unroll
        for (uint32_t k = 0; k < PPT; ++k) {
            const unsigned long long cov = __ballot(covered[k]);
            if (!cov) continue;
            uint32_t lead_s;
            const uint32_t lead = uniform_pick(it[k], cov, lead_s);
            if (__ballot(covered[k] && it[k] != lead) == 0ull) {
                // every covered pixel of the wave belongs to one draw item
                const DrawItem d = load_item_uniform(fb.items, lead_s);
                if (covered[k]) prepare_pixel<kShadows, kStored>(d, fc, fb.clip, fb.textures, s_lut, id[k], pi[k], pj[k], fresh[k], fb.counters);
            } else if (covered[k]) {
                // waterfall: one pass per distinct draw item among the wave's pixels
                for (unsigned long long pending = __ballot(true); pending;) {
                    uint32_t cur_s, cur = uniform_pick(it[k], pending, cur_s);
                    const bool mine = it[k] == cur;
                    pending &= ~__ballot(mine);
                    if (mine) {
                        const DrawItem d = load_item_uniform(fb.items, cur_s);
                        prepare_pixel<kShadows, kStored>(d, fc, fb.clip, fb.textures, s_lut, id[k], pi[k], pj[k], fresh[k], fb.counters);
                    }
                }
            }
        }
    }
    // phase B + C, per wave: the bounding box of the wave's lit fragments (4 rows of the
    // tile), then for each group of 64 lights one lane tests one light against it and the wave
    // walks the surviving bits in ascending order.  A point light whose sphere misses the box adds
    // exactly zero to every fragment in it (distance > radius discards), so dropping it changes
    // nothing, and the lights keep their original order.
#if RND_FAST_SHADE
    // Straight-line from here to the light loop: a lane without a fragment computes on whatever `fresh` holds for it
    // (zeros, or the item's first triangle) and its results are dropped by the selects behind the loop — defaults
    // written up front made every merge of the paths above re-materialize a dozen registers per pixel.
    PixelPair pp;
    {
#pragma clang fp contract(fast)
#define RND_PAIR(f) pp.f = (f32x2){fresh[0].f, fresh[1].f}
        RND_PAIR(fx); RND_PAIR(fy); RND_PAIR(fz); RND_PAIR(Nx); RND_PAIR(Ny); RND_PAIR(Nz); RND_PAIR(br); RND_PAIR(bg); RND_PAIR(bb);
#undef RND_PAIR
        pp.lit0 = covered[0] && fresh[0].lit; pp.lit1 = covered[1] && fresh[1].lit;
        // unit vector to the camera
        const f32x2 vx = splat2(fc.cam_pos[0]) - pp.fx, vy = splat2(fc.cam_pos[1]) - pp.fy, vz = splat2(fc.cam_pos[2]) - pp.fz;
        const f32x2 inv_v = rsq2(vx * vx + vy * vy + vz * vz);
        pp.Vx = vx * inv_v; pp.Vy = vy * inv_v; pp.Vz = vz * inv_v;
        pp.ar = pp.ag = pp.ab = splat2(0.0f);
        pp.NV = pp.Nx * pp.Vx + pp.Ny * pp.Vy + pp.Nz * pp.Vz;
    }
    const bool any_lit = __ballot(pp.lit0 || pp.lit1) != 0ull;
    RND_PROF(5);
    if (any_lit) {
        if (pp.lit0) { lo[0] = hi[0] = pp.fx.x; lo[1] = hi[1] = pp.fy.x; lo[2] = hi[2] = pp.fz.x; }
        if (pp.lit1) {
            lo[0] = fminf(lo[0], pp.fx.y); lo[1] = fminf(lo[1], pp.fy.y); lo[2] = fminf(lo[2], pp.fz.y);
            hi[0] = fmaxf(hi[0], pp.fx.y); hi[1] = fmaxf(hi[1], pp.fy.y); hi[2] = fmaxf(hi[2], pp.fz.y);
        }
        wave_box(lo, hi);
        RND_PROF(6);
        // every group's verdicts first (the box dies here), then the walk
        unsigned long long km_g[kMaxLights / 64], dm_g[kMaxLights / 64];
#pragma unroll
        for (uint32_t g = 0; g < kMaxLights / 64; ++g) {
            km_g[g] = 0ull; dm_g[g] = 0ull;
            if (g * 64u < n_lights) {                               // (uniform)
                // one lane, one light: kept when it is directional or its sphere reaches the box
                bool keep_light = false, directional = false;
                if (g * 64u + lane < n_lights) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(s_light + g * 64u + lane);
                    directional = a.w < 0.0f;
                    float dd = 0.0f;
                    const float pos[3] = {a.x, a.y, a.z};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float e = fmaxf(fmaxf(lo[c] - pos[c], pos[c] - hi[c]), 0.0f);
                        dd += e * e;
                    }
                    float rr = a.w * 1.001f + 1.0e-6f;      // conservative: never drop a light that could pass
                    keep_light = directional || dd <= rr * rr;
                }
                km_g[g] = __ballot(keep_light); dm_g[g] = __ballot(directional);
            }
        }
        static_assert(kMaxLights / 64 == 4, "four groups of 64 lights");
        RND_PROF(7);
        for (uint32_t g = 0; g * 64u < n_lights; ++g) {
            const unsigned long long dm = g == 0u ? dm_g[0] : g == 1u ? dm_g[1] : g == 2u ? dm_g[2] : dm_g[3];
            unsigned long long km = g == 0u ? km_g[0] : g == 1u ? km_g[1] : g == 2u ? km_g[2] : km_g[3];
            if (km == 0ull) continue;
            // the light's 32 bytes from LDS, all lanes at one address: its fields arrive in VGPRs.
            for (; km; km &= km - 1ull) {
                const uint32_t bit = (uint32_t)(__ffsll((long long)km) - 1);
                const uint32_t li = __builtin_amdgcn_readfirstlane(g * 64u + bit);
                const f32x4 a = reinterpret_cast<const f32x4*>(s_light + li)[0], b = reinterpret_cast<const f32x4*>(s_light + li)[1];
                if ((dm >> bit) & 1ull) {
                    int32_t set = -1;
                    if constexpr (kShadows) {
                        typedef const __attribute__((address_space(4))) GpuLight* cptr_light;
                        set = ((cptr_light)fb.lights + li)->shadow_index;
                    }
                    add_directional_pair<kShadows>(a, b, set, pp, fb.shadow_sets, s_sets, fc);
                } else add_point_pair(a, b, pp);
            }
        }
    }
    RND_PROF(8);
    {
        // colour = base * (ambient + lights' sum) for a lit fragment, base for an unlit one (render_mesh.wgsl:97-106),
        // else what the pixel keeps: the clear colour, or (MSAA) the colour the sample was shaded to before
#pragma clang fp contract(fast)
        const float ar = fc.ambient[0] * fc.ambient[3], ag = fc.ambient[1] * fc.ambient[3], ab = fc.ambient[2] * fc.ambient[3];
        const float tr[2] = {pp.lit0 ? ar + pp.ar.x : 1.0f, pp.lit1 ? ar + pp.ar.y : 1.0f};
        const float tg[2] = {pp.lit0 ? ag + pp.ag.x : 1.0f, pp.lit1 ? ag + pp.ag.y : 1.0f};
        const float tb[2] = {pp.lit0 ? ab + pp.ab.x : 1.0f, pp.lit1 ? ab + pp.ab.y : 1.0f};
        const float br[2] = {pp.br.x, pp.br.y}, bg[2] = {pp.bg.x, pp.bg.y}, bb[2] = {pp.bb.x, pp.bb.y};
#pragma unroll
        for (uint32_t k = 0; k < PPT; ++k) {
            const bool again = kSamples > 1 && reuse[k];
            px[k].cr = covered[k] ? br[k] * tr[k] : (again ? keep[k][0] : fc.clear_hdr[0]);
            px[k].cg = covered[k] ? bg[k] * tg[k] : (again ? keep[k][1] : fc.clear_hdr[1]);
            px[k].cb = covered[k] ? bb[k] * tb[k] : (again ? keep[k][2] : fc.clear_hdr[2]);
        }
    }
#else
#pragma unroll
    for (uint32_t k = 0; k < PPT; ++k) {
        // (every field is set here on every path: of a sample pair only the three colours in `keep` live on to the next)
        if (covered[k]) { px[k] = fresh[k]; complete_inputs(px[k], fc); }
        else {
            px[k] = PixelInputs{};
            const bool again = kSamples > 1 && reuse[k];
            px[k].cr = again ? keep[k][0] : fc.clear_hdr[0]; px[k].cg = again ? keep[k][1] : fc.clear_hdr[1];
            px[k].cb = again ? keep[k][2] : fc.clear_hdr[2];
        }
        if (covered[k] && px[k].lit) {
            lo[0] = fminf(lo[0], px[k].fx); lo[1] = fminf(lo[1], px[k].fy); lo[2] = fminf(lo[2], px[k].fz);
            hi[0] = fmaxf(hi[0], px[k].fx); hi[1] = fmaxf(hi[1], px[k].fy); hi[2] = fmaxf(hi[2], px[k].fz);
        }
    }
    wave_box(lo, hi);
    if (lo[0] <= hi[0]) {
        for (uint32_t group = 0; group < n_lights; group += 64u) {
            bool keep_light = false;
            if (group + lane < n_lights) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(s_light + group + lane);
                float dd = 0.0f;
                const float pos[3] = {a.x, a.y, a.z};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float e = fmaxf(fmaxf(lo[c] - pos[c], pos[c] - hi[c]), 0.0f);
                    dd += e * e;
                }
                float rr = a.w * 1.001f + 1.0e-6f;
                keep_light = a.w < 0.0f || dd <= rr * rr;
            }
            for (unsigned long long km = __ballot(keep_light); km; km &= km - 1ull) {
                const uint32_t li = __builtin_amdgcn_readfirstlane(group + (uint32_t)(__ffsll((long long)km) - 1));
                const GpuLight L = load_light_uniform(fb.lights, li);
#pragma unroll
                for (uint32_t k = 0; k < PPT; ++k)
                    if (px[k].lit && (kSamples == 1 || covered[k])) add_light<kShadows>(L, px[k], fb.shadow_sets, fc);
            }
        }
    }
#endif
    if constexpr (kSamples > 1) {
#pragma unroll
        for (uint32_t k = 0; k < PPT; ++k) {
            keep[k][0] = px[k].cr; keep[k][1] = px[k].cg; keep[k][2] = px[k].cb;
            sum_r = sum_r + round_f16(px[k].cr); sum_g = sum_g + round_f16(px[k].cg); sum_b = sum_b + round_f16(px[k].cb);
        }
    }
  }
    if constexpr (kSamples > 1) {
        px[0].cr = sum_r * 0.25f; px[0].cg = sum_g * 0.25f; px[0].cb = sum_b * 0.25f;
        key[0] = key0;      // the packed frame carries sample 0's depth
    }
    // phase D: f16 target, tone map, sRGB (show_texture.wgsl + the sRGB surface)
#pragma unroll
    for (uint32_t k = 0; k < kPix; ++k) {
        uint32_t local = part + wave * (64u * kPix) + k * 64u + lane;
        uint32_t i = tile_x0 + (local & (kTile - 1)), j = tile_y0 + (local / kTile);
        bool inside = i < fb.dims.width && j < fb.dims.height;
        float hr = round_f16(px[k].cr), hg = round_f16(px[k].cg), hb = round_f16(px[k].cb);
        float r = hr, g = hg, b = hb;
        if (fc.tone == RND_TONE_MAPPING_REINHARD) {
            float lum = (0.2126f * r + 0.7152f * g) + 0.0722f * b;
            float inv = rcp_f(lum + 1.0f);
#if RND_FAST_SHADE
            r = r * inv; g = g * inv; b = b * inv;
#else
            float dd = lum + 1.0f; (void)inv;
            r = r / dd; g = g / dd; b = b / dd;
#endif
        }
        uint32_t rgba = srgb8_f(r) | (srgb8_f(g) << 8) | (srgb8_f(b) << 16) | 0xFF000000u;
        if (fb.tiles_rgba8) reinterpret_cast<uint32_t*>(fb.tiles_rgba8)[(size_t)compact_s * kTilePixels + local] = inside ? rgba : 0u;
        if (inside) {
            size_t p = (size_t)j * fb.dims.width + i;
            if (fb.frame_rgba8) reinterpret_cast<uint32_t*>(fb.frame_rgba8)[p] = rgba;
            if (fb.packed) reinterpret_cast<unsigned long long*>(fb.packed)[p] = (key[k] & 0xFFFFFFFF00000000ull) | rgba;
            if (fb.hdr) { fb.hdr[p * 3] = hr; fb.hdr[p * 3 + 1] = hg; fb.hdr[p * 3 + 2] = hb; }
        }
    }
    RND_PROF(9);
  }
    RND_PROF_FLUSH;
}

// ------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------
__global__ void k_extract(const unsigned long long* __restrict__ vis, const TileRef* __restrict__ order, FrameConst fc,
                          float* depth, uint32_t* ids, uint8_t* coverage) {
    const TileRef ref = order[blockIdx.x];
    const uint32_t tile_x0 = (ref.tile % fc.tiles_x) * kTile, tile_y0 = (ref.tile / fc.tiles_x) * kTile;
    for (uint32_t local = threadIdx.x; local < kTilePixels; local += blockDim.x) {
        uint32_t i = tile_x0 + (local & (kTile - 1)), j = tile_y0 + local / kTile;
        if (i >= fc.width || j >= fc.height) continue;
        unsigned long long key = vis[(size_t)ref.tile * kTilePixels + local];
        size_t p = (size_t)j * fc.width + i;
        uint32_t id = (uint32_t)(key & 0xFFFFFFFFull);
        if (depth) depth[p] = __uint_as_float((uint32_t)(key >> 32));
        if (ids) ids[p] = id;
        if (coverage) coverage[p] = id != RND_NONE;
    }
}

__global__ void k_split_positions(const rnd_vertex* __restrict__ verts, uint32_t n, float* px, float* py, float* pz) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    px[i] = verts[i].position[0];
    py[i] = verts[i].position[1];
    pz[i] = verts[i].position[2];
}

__global__ void k_detile(const uint32_t* __restrict__ tiles_rgba8, const uint32_t* __restrict__ tiles, FrameConst fc,
                         uint32_t* frame) {
    const uint32_t tile = tiles[blockIdx.x];
    const uint32_t tile_x0 = (tile % fc.tiles_x) * kTile, tile_y0 = (tile / fc.tiles_x) * kTile;
    for (uint32_t local = threadIdx.x; local < kTilePixels; local += blockDim.x) {
        uint32_t i = tile_x0 + (local & (kTile - 1)), j = tile_y0 + local / kTile;
        if (i < fc.width && j < fc.height) frame[(size_t)j * fc.width + i] = tiles_rgba8[(size_t)blockIdx.x * kTilePixels + local];
    }
}

// All ranks' tile-major buffers (rank r at gathered + r * stride_words) into the row-major frame in one
// launch, one workgroup per tile of the frame.  With full-width stripes of kOwnerGroup tile rows dealt
// round-robin from the last rank, tile (tx, ty) belongs to rank stripe_owner(ty / kOwnerGroup, world) and is number
// ((ty / kOwnerGroup / world) * kOwnerGroup + ty % kOwnerGroup) * tiles_x + tx of that rank's ascending list.
__global__ void k_detile_gathered(const uint32_t* __restrict__ gathered, uint64_t stride_words, uint32_t world,
                                  FrameConst fc, uint32_t* frame) {
    const uint32_t tile = blockIdx.x, tx = tile % fc.tiles_x, ty = tile / fc.tiles_x;
    const uint32_t group = ty / kOwnerGroup, owner = stripe_owner(group, world);
    const uint32_t compact = ((group / world) * kOwnerGroup + ty % kOwnerGroup) * fc.tiles_x + tx;
    const uint32_t* src = gathered + (size_t)owner * stride_words + (size_t)compact * kTilePixels;
    const uint32_t tile_x0 = tx * kTile, tile_y0 = ty * kTile;
    for (uint32_t local = threadIdx.x; local < kTilePixels; local += blockDim.x) {
        uint32_t i = tile_x0 + (local & (kTile - 1)), j = tile_y0 + local / kTile;
        if (i < fc.width && j < fc.height) frame[(size_t)j * fc.width + i] = src[local];
    }
}

// ------------------------------------------------------------------------------------------
// 2D overlay (raster rule 14; pipeline2d.rs:266-310, backend.rs:470-491).  One workgroup per 32x32
// tile of the surface, four pixels per lane.  Per pixel the Rgba8UnormSrgb 2D target starts
// transparent and takes, in draw order, the debug view, the boxes and the glyphs that cover the
// pixel centre, each with ALPHA_BLENDING and an 8-bit store; the result goes over the presented
// frame with the same blend.  Quads are axis-aligned, so coverage is a half-open interval test on
// the snapped corners.  The quads of a batch of 256 that touch the tile are compacted, in order,
// into LDS (ballot + prefix) before the pixels walk them.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sat_f(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float smoothstep_f(float e0, float e1, float x) {
    const float t = sat_f((x - e0) / (e1 - e0));
    return (t * t) * (3.0f - 2.0f * t);
}
__device__ __forceinline__ uint32_t blend_srgb8(uint32_t d, float r, float g, float b, float a_in, const float* __restrict__ lut) {
    const float a = sat_f(a_in), ia = 1.0f - a;
    const uint32_t o0 = srgb8(sat_f(r) * a + lut[d & 255u] * ia);
    const uint32_t o1 = srgb8(sat_f(g) * a + lut[(d >> 8) & 255u] * ia);
    const uint32_t o2 = srgb8(sat_f(b) * a + lut[(d >> 16) & 255u] * ia);
    const float oa = a * 1.0f + ((float)(d >> 24) / 255.0f) * ia;
    const uint32_t o3 = (uint32_t)(int)(sat_f(oa) * 255.0f + 0.5f);
    return o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
}

__global__ __launch_bounds__(256) void k_overlay(OverlayParams p) {
    __shared__ float s_lut[256];
    __shared__ Prim2D s_prims[256];
    __shared__ uint32_t s_wave_count[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
    if (p.world > 1u && !owns_tile(tx, ty, p.rank, p.world)) return;
    // a truncated frame was not shaded: the image still holds the frame before it WITH its overlay; blending again would
    // composite translucent boxes and glyph edges twice
    if (frame_overflowed(p.gate)) return;
    s_lut[tid] = p.srgb_lut[tid];
    const int32_t tile_x0 = (int32_t)(tx * kTile), tile_y0 = (int32_t)(ty * kTile);
    // sub-pixel bounds of the tile's pixel centres
    const int32_t cx_lo = tile_x0 * kSub + kSub / 2, cx_hi = (tile_x0 + (int32_t)kTile - 1) * kSub + kSub / 2;
    const int32_t cy_lo = tile_y0 * kSub + kSub / 2, cy_hi = (tile_y0 + (int32_t)kTile - 1) * kSub + kSub / 2;
    uint32_t dst[4] = {0u, 0u, 0u, 0u};
    __syncthreads();
    if (p.fullscreen) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t local = k * 256u + tid;
            const uint32_t i = (uint32_t)tile_x0 + (local & (kTile - 1)), j = (uint32_t)tile_y0 + local / kTile;
            if (i >= p.sw || j >= p.sh) continue;
            float depth = 1.0f;
            if (p.depth_src) {
                const float u = ((float)i + 0.5f) / (float)p.sw, v = ((float)j + 0.5f) / (float)p.sh;
                const uint32_t x = min((uint32_t)floorf(u * (float)p.src_w), p.src_w - 1u), y = min((uint32_t)floorf(v * (float)p.src_h), p.src_h - 1u);
                const size_t at = (size_t)((y / kTile) * p.src_tiles_x + x / kTile) * kTilePixels + ((y % kTile) * kTile + x % kTile);
                depth = __uint_as_float((uint32_t)(p.depth_src[at] >> 32));
            }
            const float dd = (2.0f * 0.5f) / ((50.0f + 0.5f) - depth * (50.0f - 0.5f));
            dst[k] = blend_srgb8(dst[k], dd, dd, dd, 1.0f, s_lut);
        }
    }
    for (uint32_t base = 0; base < p.n_prims; base += 256u) {
        // this batch's quads that are front-facing and reach the tile, kept in order
        bool keep = false;
        Prim2D q;
        if (base + tid < p.n_prims) {
            q = p.prims[base + tid];
            const long long area = (long long)(q.xb - q.xa) * (long long)(q.yb - q.yt);
            const int32_t x_lo = min(q.xa, q.xb), x_hi = max(q.xa, q.xb), y_lo = min(q.yt, q.yb), y_hi = max(q.yt, q.yb);
            keep = area > 0 && x_lo <= cx_hi && x_hi > cx_lo && y_lo <= cy_hi && y_hi > cy_lo;
        }
        const unsigned long long km = __ballot(keep);
        if (lane == 0) s_wave_count[wave] = (uint32_t)__popcll(km);
        __syncthreads();
        uint32_t off = 0, total = 0;
        for (uint32_t w = 0; w < 4; ++w) { const uint32_t c = s_wave_count[w]; if (w < wave) off += c; total += c; }
        if (keep) s_prims[off + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = q;
        __syncthreads();
        for (uint32_t n = 0; n < total; ++n) {
            const Prim2D& r = s_prims[n];
            const int32_t x_lo = min(r.xa, r.xb), x_hi = max(r.xa, r.xb), y_lo = min(r.yt, r.yb), y_hi = max(r.yt, r.yb);
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t local = k * 256u + tid;
                const int32_t px = (tile_x0 + (int32_t)(local & (kTile - 1))) * kSub + kSub / 2;
                const int32_t py = (tile_y0 + (int32_t)(local / kTile)) * kSub + kSub / 2;
                if (px < x_lo || px >= x_hi || py < y_lo || py >= y_hi) continue;
                if (r.kind == 0u) {
                    dst[k] = blend_srgb8(dst[k], r.color[0], r.color[1], r.color[2], r.color[3], s_lut);
                } else {
                    const float u = (float)(px - r.xa) / (float)(r.xb - r.xa), v = (float)(py - r.yt) / (float)(r.yb - r.yt);
                    const float cell = 1.0f / 128.0f;
                    const float u0 = (0.0f / 128.0f) + (float)r.glyph * cell, u1 = (1.0f / 128.0f) + (float)r.glyph * cell;
                    const float tu = u0 + (u1 - u0) * u;
                    const f32x4 sd = texture_bilinear(p.font, 0u, tu, v, FetchF32{(gptr_f32x4)p.font.texels});
                    const float ratio = (r.font_size - 12.0f) / (24.0f - 12.0f);
                    const float smoothing = fminf(fmaxf(0.25f + ratio * (0.15f - 0.25f), 0.1f), 0.9f);
                    const float fill_dist = smoothstep_f(0.37f - smoothing, 0.37f + smoothing, sd.z);
                    const float outline_dist = smoothstep_f(0.8f - smoothing, 0.8f + smoothing, sd.z);
                    const float fill_alpha = 1.0f - fill_dist, alpha = 1.0f - outline_dist;
                    const float col = 0.0f * (1.0f - fill_alpha) + 1.0f * fill_alpha;
                    dst[k] = blend_srgb8(dst[k], col, col, col, alpha, s_lut);
                }
            }
        }
        __syncthreads();
    }
    // out: the 2D target, and the 2D target over the presented frame
    uint32_t compact = 0;
    if (p.world > 1u) compact = ((ty / kOwnerGroup / p.world) * kOwnerGroup + ty % kOwnerGroup) * p.tiles_x + tx;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t local = k * 256u + tid;
        const uint32_t i = (uint32_t)tile_x0 + (local & (kTile - 1)), j = (uint32_t)tile_y0 + local / kTile;
        if (i >= p.sw || j >= p.sh) continue;
        const size_t at = (size_t)j * p.sw + i;
        if (p.overlay_out) p.overlay_out[at] = dst[k];
        if ((dst[k] >> 24) == 0u) continue;
        uint32_t* f = p.frame ? p.frame + at : p.tiles + (size_t)compact * kTilePixels + local;
        const uint32_t d = dst[k], old = *f;
        const float a = (float)(d >> 24) / 255.0f, ia = 1.0f - a;
        const uint32_t o0 = srgb8(s_lut[d & 255u] * a + s_lut[old & 255u] * ia);
        const uint32_t o1 = srgb8(s_lut[(d >> 8) & 255u] * a + s_lut[(old >> 8) & 255u] * ia);
        const uint32_t o2 = srgb8(s_lut[(d >> 16) & 255u] * a + s_lut[(old >> 16) & 255u] * ia);
        *f = o0 | (o1 << 8) | (o2 << 16) | (old & 0xFF000000u);
    }
}

__global__ void k_present(const float* __restrict__ hdr, uint32_t w, uint32_t h, uint32_t* __restrict__ frame,
                          uint32_t sw, uint32_t sh, uint32_t linear, uint32_t tone, OverflowGate gate) {
    if (frame_overflowed(gate)) return;          // (the surface keeps the frame before, overlay included)
    const uint32_t i = blockIdx.x * 64u + (threadIdx.x & 63u), j = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (i >= sw || j >= sh) return;
    const float u = ((float)i + 0.5f) / (float)sw, v = ((float)j + 0.5f) / (float)sh;
    float c[3];
    if (!linear) {          // unfiltered sampler: nearest, ClampToEdge (backend.rs:428-439)
        const uint32_t x = min((uint32_t)floorf(u * (float)w), w - 1u), y = min((uint32_t)floorf(v * (float)h), h - 1u);
        const float* p = hdr + ((size_t)y * w + x) * 3;
        c[0] = p[0]; c[1] = p[1]; c[2] = p[2];
    } else {                // filtered sampler: linear, Repeat (backend.rs:414-426); one level, isotropic
        const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
        const float xf = floorf(x), yf = floorf(y), fx = x - xf, fy = y - yf;
        auto wrap = [](float f, uint32_t n) { long long k = (long long)f % (long long)n; return (uint32_t)(k < 0 ? k + (long long)n : k); };
        const uint32_t x0 = wrap(xf, w), x1 = wrap(xf + 1.0f, w), y0 = wrap(yf, h), y1 = wrap(yf + 1.0f, h);
        const float* p00 = hdr + ((size_t)y0 * w + x0) * 3, *p10 = hdr + ((size_t)y0 * w + x1) * 3;
        const float* p01 = hdr + ((size_t)y1 * w + x0) * 3, *p11 = hdr + ((size_t)y1 * w + x1) * 3;
        for (int k = 0; k < 3; ++k) {
            const float top = p00[k] * (1.0f - fx) + p10[k] * fx, bot = p01[k] * (1.0f - fx) + p11[k] * fx;
            c[k] = top * (1.0f - fy) + bot * fy;
        }
    }
    float r = c[0], g = c[1], b = c[2];
    if (tone == RND_TONE_MAPPING_REINHARD) {
        const float lum = (0.2126f * r + 0.7152f * g) + 0.0722f * b;
        const float dd = lum + 1.0f;
        r = r / dd; g = g / dd; b = b / dd;
    }
    frame[(size_t)j * sw + i] = srgb8(r) | (srgb8(g) << 8) | (srgb8(b) << 16) | 0xFF000000u;
}

__global__ void k_unpack(const unsigned long long* __restrict__ packed, FrameConst fc, uint32_t* frame, float* depth) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)fc.width * fc.height) return;
    unsigned long long v = packed[p];
    if (frame) frame[p] = (uint32_t)(v & 0xFFFFFFFFull);
    if (depth) depth[p] = __uint_as_float((uint32_t)(v >> 32));
}

// Sort-last depth composite of one strip: element-wise minimum of `n` packed (depth bits << 32 | rgba8) strips,
// strip r at strips + r * stride (the layout an all-to-all leaves on the rank that owns the strip).  Unsigned
// compare = depth compare (depth in [0, 1]); equal depths resolve by the colour bits.  Two words per lane per load.
__global__ __launch_bounds__(256) void k_min_packed(const unsigned long long* __restrict__ strips, uint32_t n, uint64_t stride,
                                                    unsigned long long* __restrict__ out, uint64_t count) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const uint64_t pairs = count / 2u;
    for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < pairs; k += (uint64_t)gridDim.x * 256u) {
        u64x2 m = reinterpret_cast<const u64x2*>(strips)[k];
        for (uint32_t r = 1; r < n; ++r) {
            const u64x2 v = reinterpret_cast<const u64x2*>(strips + r * stride)[k];
            m.x = v.x < m.x ? v.x : m.x; m.y = v.y < m.y ? v.y : m.y;
        }
        reinterpret_cast<u64x2*>(out)[k] = m;
    }
    if ((count & 1u) && blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long m = strips[count - 1u];
        for (uint32_t r = 1; r < n; ++r) { const unsigned long long v = strips[r * stride + count - 1u]; m = v < m ? v : m; }
        out[count - 1u] = m;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline uint32_t div_up(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

void launch_clear(const FrameBuffers& fb, const FrameConst& fc, hipStream_t stream, const void* host_tables,
                  void* dev_tables, size_t table_bytes) {
    const uint32_t n_counts = 2u * fc.tiles_x * fc.tiles_y;
    // (an early-depth frame in two passes has a second row of vertex-block flags: the far units', set by k_cull_far)
    const uint32_t n_vblocks = (fc.n_verts + kVertexBlock - 1) / kVertexBlock * std::max(fc.n_views, 1u) * (fb.early_z && fb.setup_pass ? 2u : 1u);
    const uint32_t n_vec = (uint32_t)(table_bytes / 16);
    // enough workgroups for the table copy even when this context owns few tiles
    const uint32_t blocks = std::max<uint32_t>(fb.n_owned ? fb.n_owned : 1u, std::min<uint32_t>(div_up(n_vec, 256), 1024u));
    hipLaunchKernelGGL(k_clear, dim3(blocks), dim3(256), 0, stream, fb, n_counts, n_vblocks, std::max(fc.n_samples, 1u),
                       static_cast<const uint4*>(host_tables), static_cast<uint4*>(dev_tables), n_vec);
}

void launch_cull(const FrameBuffers& fb, const FrameConst& fc, hipStream_t stream) {
    if (fc.n_units) hipLaunchKernelGGL(k_cull, dim3(div_up(fc.n_units, 256)), dim3(256), 0, stream, fb);
}

void launch_transform(const FrameBuffers& fb, const FrameConst& fc, hipStream_t stream) {
    if (!fc.n_verts) return;
    hipLaunchKernelGGL(k_transform<1>, dim3(div_up(fc.n_verts, 256), std::max(fc.n_views, 1u)), dim3(256), 0, stream, fb.items, fb.vert_bases, fb.frame_const, fb.clip, fb.snap,
                       fb.vblock_needed, fb.counters);
}
void launch_setup(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream) {
    if (!fc.n_tris) return;
    const bool ms = fc.n_samples == 4u;
    const dim3 grid(fc.n_units + kXcds - 1u);
    // (experiment aid: unused dynamic LDS per workgroup lowers the occupancy of the main pass's setup kernel)
    static const unsigned lds_pad = [] { const char* e = getenv("RND_SETUP_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
    if (lds_pad && fc.n_views <= 1u && !fb.alpha_test && !ms) {
        hipLaunchKernelGGL((k_setup<false, 1, false, false>), grid, dim3(256), lds_pad, stream, fb, cap);
        return;
    }
    if (fc.n_views > 1u) {
        // a light's cascades: depth only (no alpha test, one sample; visual_server.cpp builds them so)
        hipLaunchKernelGGL((k_setup<false, 1, true, false>), grid, dim3(256), 0, stream, fb, cap);
    } else if (fb.alpha_test) {
        if (ms) hipLaunchKernelGGL((k_setup<true, 4, false, false>), grid, dim3(256), 0, stream, fb, cap);
        else hipLaunchKernelGGL((k_setup<true, 1, false, false>), grid, dim3(256), 0, stream, fb, cap);
    } else {
        // (the early-depth variant exists for the plain main pass only: the frames it is for are that)
        if (ms) hipLaunchKernelGGL((k_setup<false, 4, false, false>), grid, dim3(256), 0, stream, fb, cap);
        else if (fb.early_z) {
            if (fb.setup_pass == 0u) hipLaunchKernelGGL((k_setup<false, 1, false, true>), grid, dim3(256), 0, stream, fb, cap);
            else {
                FrameBuffers pass = fb;
                pass.setup_pass = 1u;
                hipLaunchKernelGGL((k_setup<false, 1, false, true>), grid, dim3(256), 0, stream, pass, cap);
                hipLaunchKernelGGL(k_hiz, dim3(fc.tiles_x * fc.tiles_y), dim3(256), 0, stream, fb);
                hipLaunchKernelGGL(k_cull_far, dim3(div_up(fc.n_units, 4)), dim3(256), 0, stream, fb);
                hipLaunchKernelGGL(k_transform<8>, dim3(div_up(div_up(fc.n_verts, 256), 8), 1u), dim3(256), 0, stream, fb.items, fb.vert_bases, fb.frame_const, fb.clip, fb.snap,
                                   fb.vblock_needed + div_up(fc.n_verts, 256), fb.counters);
                pass.setup_pass = 2u;
                hipLaunchKernelGGL((k_setup<false, 1, false, true>), grid, dim3(256), 0, stream, pass, cap);
            }
        }
        else hipLaunchKernelGGL((k_setup<false, 1, false, false>), grid, dim3(256), 0, stream, fb, cap);
    }
}
void launch_clip(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream) {
    if (!fc.n_tris) return;
    uint32_t blocks = min(div_up(cap.clip_queue, 256), 256u);       // grid-stride over the queue; it is empty in most frames
    hipLaunchKernelGGL(k_clip, dim3(blocks), dim3(256), 0, stream, fb, cap);
}
void launch_scan(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, stream, fb, cap);
}
void launch_bin(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream) {
    if (!fc.n_tris) return;
    uint32_t blocks = fc.n_units + kBinClipBlocks;
    hipLaunchKernelGGL(k_bin_fill, dim3(blocks), dim3(256), 0, stream, fb, cap);
}
void launch_raster(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream) {
    if (!fb.n_owned) return;
    const bool ms = fc.n_samples == 4u;
    const dim3 grid(fb.n_owned + kRasterExtraCap);
    if (fb.alpha_test) {
        if (ms) hipLaunchKernelGGL((k_raster<true, 4>), grid, dim3(256), 0, stream, fb, cap);
        else hipLaunchKernelGGL((k_raster<true, 1>), grid, dim3(256), 0, stream, fb, cap);
    } else {
        if (ms) hipLaunchKernelGGL((k_raster<false, 4>), grid, dim3(256), 0, stream, fb, cap);
        else hipLaunchKernelGGL((k_raster<false, 1>), grid, dim3(256), 0, stream, fb, cap);
    }
}
void launch_shade(const FrameBuffers& fb, const FrameConst& fc, hipStream_t stream) {
    if (!fb.n_owned) return;
    const bool sh = fb.shadow_sets != nullptr || fb.has_textures;
    const bool ms = fc.n_samples == 4;
    // Dynamic LDS: the lights (32 B each), the sRGB table, and an unused remainder that sets the occupancy.  The plain
    // variant is limited to seven workgroups a CU by its registers; claiming LDS so that LDS says seven as well made the
    // next frame's geometry kernels, which run beside this one, interleave better.
    // RND_SHADE_LDS_PAD overrides the remainder for every variant (the occupancy experiments).
    static const int lds_env = [] { const char* e = getenv("RND_SHADE_LDS_PAD"); return e ? atoi(e) : -1; }();
    // one workgroup per part (512 pixels; 256 under MSAA); part numbers interleave the XCDs: see k_shade's locate
    const uint32_t n_parts = (fb.n_owned + kXcds - 1u) / kXcds * kXcds * (ms ? 4u : 2u);
    const uint32_t grid = n_parts;
    const uint32_t tables = std::max(std::min(fc.n_lights, kMaxLights), 1u) * (uint32_t)sizeof(ShadeLight);
    const bool plain = !ms && !sh;
    const unsigned pad = lds_env >= 0 ? (unsigned)lds_env : (plain ? 1024u : 0u);
    const unsigned lds = tables + pad;
    if (ms) {
        if (sh) hipLaunchKernelGGL((fb.clip ? k_shade<true, 4, true> : k_shade<true, 4, false>), dim3(grid), dim3(256), lds, stream, fb, n_parts);
        else hipLaunchKernelGGL((fb.clip ? k_shade<false, 4, true> : k_shade<false, 4, false>), dim3(grid), dim3(256), lds, stream, fb, n_parts);
    } else {
        if (sh) hipLaunchKernelGGL((fb.clip ? k_shade<true, 1, true> : k_shade<true, 1, false>), dim3(grid), dim3(256), lds, stream, fb, n_parts);
        else hipLaunchKernelGGL((fb.clip ? k_shade<false, 1, true> : k_shade<false, 1, false>), dim3(grid), dim3(256), lds, stream, fb, n_parts);
    }
}
void launch_extract(const uint64_t* vis, const TileRef* order, uint32_t n_owned, const FrameConst& fc, float* depth,
                    uint32_t* ids, uint8_t* coverage, hipStream_t stream) {
    if (!n_owned) return;
    hipLaunchKernelGGL(k_extract, dim3(n_owned), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned long long*>(vis), order, fc, depth, ids, coverage);
}
void launch_split_positions(const rnd_vertex* verts, uint32_t n, float* px, float* py, float* pz, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_split_positions, dim3(div_up(n, 256)), dim3(256), 0, stream, verts, n, px, py, pz);
}
void launch_detile(const uint8_t* tiles_rgba8, const uint32_t* tiles, uint32_t n, const FrameConst& fc,
                   uint8_t* frame_rgba8, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_detile, dim3(n), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(tiles_rgba8), tiles,
                       fc, reinterpret_cast<uint32_t*>(frame_rgba8));
}
void launch_detile_gathered(const uint8_t* gathered, uint64_t stride_bytes, uint32_t world, const FrameConst& fc,
                            uint8_t* frame_rgba8, hipStream_t stream) {
    hipLaunchKernelGGL(k_detile_gathered, dim3(fc.tiles_x * fc.tiles_y), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t*>(gathered), stride_bytes / 4, world, fc, reinterpret_cast<uint32_t*>(frame_rgba8));
}
void launch_present(const float* hdr, uint32_t w, uint32_t h, uint8_t* frame_rgba8, uint32_t sw, uint32_t sh,
                    bool linear, uint32_t tone, const OverflowGate& gate, hipStream_t stream) {
    hipLaunchKernelGGL(k_present, dim3(div_up(sw, 64), div_up(sh, 4)), dim3(256), 0, stream, hdr, w, h,
                       reinterpret_cast<uint32_t*>(frame_rgba8), sw, sh, linear ? 1u : 0u, tone, gate);
}
void launch_overlay(const OverlayParams& p, uint32_t n_tiles, hipStream_t stream) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(k_overlay, dim3(n_tiles), dim3(256), 0, stream, p);
}
void launch_min_packed(const uint64_t* strips, uint32_t n, uint64_t stride, uint64_t* out, uint64_t count, hipStream_t stream) {
    if (!count || !n) return;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((count / 2u + 255u) / 256u + 1u, 256u * 16u);
    hipLaunchKernelGGL(k_min_packed, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const unsigned long long*>(strips), n, stride,
                       reinterpret_cast<unsigned long long*>(out), count);
}
void launch_unpack(const uint64_t* packed, const FrameConst& fc, uint8_t* frame_rgba8, float* depth, hipStream_t stream) {
    size_t n = (size_t)fc.width * fc.height;
    hipLaunchKernelGGL(k_unpack, dim3(div_up(n, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned long long*>(packed), fc, reinterpret_cast<uint32_t*>(frame_rgba8), depth);
}

}  // namespace rnd
