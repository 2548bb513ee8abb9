// Register-resident layer chains on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16):
// the same chains as chain_reg.hip - a wave carries its points through consecutive
// shared-MLP layers, every layer computed transposed so that its result is already the
// next layer's operand - with bf16 weights and bf16 activations between the layers and
// fp32 accumulation.  The first layer of every stack reads coordinates in metres, which
// bf16's 8 bits of mantissa would quantise to decimetres: it multiplies split operands
// (FirstLayer below).  Kept in fp32: the batch-norm shifts, the maxima over points and the
// final 128 -> 2 layer.
//
// Operand shapes.  One MFMA contracts over 16 channels; lane (p = lane & 31,
// h = lane >> 5) supplies 8 of them.  A 32 x 32 fp32 result tile (lane = point p,
// register r = channel (r & 3) + 8 (r >> 2) + 4 h) becomes two operand fragments by
// converting registers 8s .. 8s+7 (s = 0, 1) pairwise to bf16: element e of fragment s
// is channel 16 s + 8 (e >> 2) + 4 h + (e & 3) of the tile.  The weight fragments are
// laid out to match (fpn_chain_pack_bf16), with the batch-norm scale already multiplied
// into the columns, so that an epilogue is: add the shift, clamp at zero, convert.
//
// Weight stream.  A UNIT is what one activation tile needs against a PAIR of output
// tiles: 2 tiles x 2 fragments x 64 lanes x 16 bytes = 4 KiB.  The units of a kernel are
// stored back to back in exactly the order in which it consumes them (layer by layer;
// inside a layer pair-major, input tile minor), followed by kPadUnits units of padding,
// so that the kernel can always read a fixed number of units ahead - across layer
// boundaries - through a ring of registers.
//
// Weight delivery.  Every MFMA needs 1 KiB of weight fragments; four waves at the MFMA's
// pace, each reading the stream for itself, would pull 128 bytes per clock through the
// CU's vector memory path, which delivers 64.  So where the registers allow, a wave
// carries NP = 2 groups of 32 points and uses every fragment twice (then the weight reads
// cost 10-15 % of the kernel: seg head 0.31 ms against 0.27 ms with the loads removed and
// the fragments kept opaque to the compiler, box chain 0.17 against 0.15).  The seg tail
// cannot (conv6's output fills the registers) and takes its units through a ring in LDS
// that its workgroup fills once (CoopFeed below): 0.57 -> 0.50 ms.  With no weight
// movement at all the tail takes 0.37 ms, twice its MFMA time: one wave per SIMD issues
// in order, and every epilogue, wait and barrier is a bubble in the matrix pipe.
// (A timing build that removes loads must keep the fragments opaque - FPN_RING_NOLOAD,
// FPN_FEED_NO* below - or the compiler merges the MFMAs that now see equal operands.)
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <utility>

#include "fpn_common.h"

namespace fpn {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef short shortx2 __attribute__((ext_vector_type(2)));
typedef short shortx8 __attribute__((ext_vector_type(8)));

constexpr int kWaves = 4;          // waves per workgroup: they walk the weight stream together
constexpr int kUnitVecs = 256;     // 16-byte vectors per unit
constexpr int kPadUnits = 16;      // read-ahead beyond the last unit of a stream
constexpr int kRing = 4;           // units in flight per wave

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(std::forward<F>(f));
    }
}

// LDS executes one wave's instructions in issue order: what is needed between the ds_max updates of a row and
// the wave's own reads of it (and between those reads and the next updates) is only that the COMPILER keeps
// the order.  (A fence, even at wavefront scope, also waits for the wave's outstanding memory reads - the
// next round's inputs.)
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

struct Act {            // 32 channels of 32 points as two operand fragments
    bf16x8 m[2];
};

struct WUnit {          // [tile of the pair][fragment]
    bf16x8 w[2][2];
};

__device__ __forceinline__ void load_unit(WUnit &u, const bf16x8 *__restrict__ p, unsigned lane)
{
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m) u.w[a][m] = p[(a * 2 + m) * 64 + lane];
}

// The same through a buffer descriptor: the lane's 16-byte slot is the instruction's vector offset (one
// register for the whole kernel), the unit's position a scalar offset, the fragment an immediate - no vector
// arithmetic per load.  (Read through per-lane pointers, every unit of an unrolled chain gets a 64-bit address
// pair of its own; the compiler parks the pairs in accumulator registers and pays two moves per load - of the
// seg head's 24 k cycles per round in conv5, 5 k were the weight loads' instructions.)
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
struct StreamWindow {
    __amdgpu_buffer_rsrc_t rsrc;
    const bf16x8 *base;
    __device__ __forceinline__ void bind(const bf16x8 *ws)
    {
        base = ws;
        // (gfx9 raw buffer: 32-bit elements; the window is as long as an offset can be)
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(ws), 0, 0x7fffffff, 0x00020000);
    }
    __device__ __forceinline__ void load(WUnit &u, const bf16x8 *__restrict__ unit, unsigned lane) const
    {
        const int soff = (int)(reinterpret_cast<uintptr_t>(unit) - reinterpret_cast<uintptr_t>(base));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                u.w[a][m] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16), soff + (a * 2 + m) * 1024, 0));
    }
};

// Where a wave's weight units come from.  Both feeds hand out the unit of stream position
// PH (mod 4) and are advanced once per unit, right after the unit's MFMAs are issued.
//
// RingFeed: the wave reads the stream itself, kRing units ahead, 4 KiB per unit through
// the vector memory path - with four waves at the MFMA's pace that path (64 bytes a clock
// per CU) is the limit as soon as a fragment is used once only (NP = 1).
//
// CoopFeed: the four waves of a workgroup walk the SAME stream, so each unit crosses the
// vector memory path once per workgroup - every wave fetches a quarter of it (one 16-byte
// load per lane), parks it in registers for kFeedStage units, writes it into a ring of
// kFeedSlots units in LDS, and all four read their fragments from there (ds_read_b128,
// 256 bytes a clock per CU).  Unit x is written during step x - kFeedAhead, read into
// registers during step x - kFeedRead and multiplied in step x; a workgroup barrier stands
// before every even step, which puts one between the write and the read of a unit and one
// between the last read of a slot and its next write.  Needs every wave of the workgroup
// alive and on the same stream.
constexpr int kFeedSlots = 4, kFeedAhead = 4, kFeedRead = 2, kFeedStage = 4;
static_assert(kFeedAhead + kFeedStage < kPadUnits, "read-ahead exceeds the stream's padding");

// (timing-only builds) the compiler may assume nothing about the unit's registers
__device__ __forceinline__ void opaque(WUnit &u)
{
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m) asm volatile("" : "+v"(u.w[a][m]));
}

// DEPTH units in flight (= the period of the slot numbers PH)
template <int DEPTH>
struct RingFeedT {
    static constexpr int kMode = 0, kDepth = DEPTH;
    WUnit ring[DEPTH];
    unsigned lane;
    StreamWindow win;
    __device__ __forceinline__ void start(const bf16x8 *__restrict__ ws, unsigned lane_, bf16x8 *)
    {
        lane = lane_;
        win.bind(ws);
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) win.load(ring[i], ws + (size_t)i * kUnitVecs, lane);
    }
    template <int PH>
    __device__ __forceinline__ const WUnit &cur() const { return ring[PH % DEPTH]; }
    // A kernel that does not consume its stream front to back (the seg tail interleaves two
    // layers) names the units itself: kLook = how many consumption steps ahead a feed fetches,
    // start_at(first(i)) = the unit consumed i-th, advance_at(next) = the unit consumed kLook
    // steps after the one being multiplied.
    static constexpr int kLook = DEPTH;
    // (ws: the stream's first unit - no unit named later lies before it)
    template <class First>
    __device__ __forceinline__ void start_at(First &&first, unsigned lane_, bf16x8 *, const bf16x8 *__restrict__ ws)
    {
        lane = lane_;
        win.bind(ws);
        static_for<0, DEPTH>([&](auto I) { win.load(ring[decltype(I)::value], first(I), lane); });
    }
    template <int PH>
    __device__ __forceinline__ void advance_at(const bf16x8 *__restrict__ next)
    {
#ifndef FPN_RING_NOLOAD
        win.load(ring[PH % DEPTH], next, lane);
#else       // timing-only build (wrong results): no loads; the fragments stay opaque so that every MFMA remains
        opaque(ring[PH % DEPTH]);
#endif
    }
    // unit: the unit being multiplied; its slot is refilled with the unit DEPTH further on
    template <int PH>
    __device__ __forceinline__ void advance(const bf16x8 *__restrict__ unit)
    {
        advance_at<PH>(unit + (size_t)DEPTH * kUnitVecs);
    }
    template <int PH>
    __device__ __forceinline__ void sync() {}
};
using RingFeed = RingFeedT<kRing>;

// The ring of a kernel that walks its stream again and again (one round of points after another): past the
// stream's last unit the read-ahead goes on at its first unit, so a round ends with the next round's first
// units in the ring.  The stream's unit count must be a multiple of the ring's depth (the slot numbers come
// round).  end = the first vector behind the last unit, len = the stream's length in vectors.
template <int DEPTH>
struct WrapRingFeedT : RingFeedT<DEPTH> {
    const bf16x8 *end;
    long len;
    template <int PH>
    __device__ __forceinline__ void advance(const bf16x8 *__restrict__ unit)
    {
        const bf16x8 *next = unit + (size_t)DEPTH * kUnitVecs;
        this->template advance_at<PH>(next >= end ? next - len : next);
    }
};

// The seg tail's ring when a wave walks a frustum round after round: the kernel consumes the first units of its
// stream out of stream order (TailOrder::pos), so the read-ahead past the stream's last unit goes on at the units
// the next round consumes first.  first4[i] = stream position of the unit consumed i-th.
template <int DEPTH>
struct TailWrapFeedT : RingFeedT<DEPTH> {
    const bf16x8 *base, *end;
    int first4[DEPTH];
    template <int PH>
    __device__ __forceinline__ void advance(const bf16x8 *__restrict__ unit)
    {
        const bf16x8 *next = unit + (size_t)DEPTH * kUnitVecs;
        if (next >= end) {
            const int i = (int)((next - end) / kUnitVecs);
            int pos = first4[0];
#pragma unroll
            for (int k = 1; k < DEPTH; ++k) pos = i == k ? first4[k] : pos;
            next = base + (size_t)pos * kUnitVecs;
        }
        this->template advance_at<PH>(next);
    }
};

struct CoopFeed {
    static constexpr int kMode = 1, kDepth = 4;
    WUnit frag[4];          // (indexed by stream position mod 4; kFeedRead + 1 of them are live)
    bf16x8 stage[kFeedStage];
    bf16x8 *lds;            // kFeedSlots * kUnitVecs vectors, shared by the workgroup
    unsigned lane, mine;    // mine: this lane's vector of a unit's quarter
    template <int SLOT>
    __device__ __forceinline__ void read_unit(WUnit &u) const
    {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int m = 0; m < 2; ++m) u.w[a][m] = lds[SLOT * kUnitVecs + (a * 2 + m) * 64 + lane];
    }
    __device__ __forceinline__ void read_unit_rt(WUnit &u, int slot) const
    {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int m = 0; m < 2; ++m) u.w[a][m] = lds[slot * kUnitVecs + (a * 2 + m) * 64 + lane];
    }
    __device__ __forceinline__ void start(const bf16x8 *__restrict__ ws, unsigned lane_, bf16x8 *ring)
    {
        lane = lane_;
        lds = ring;
        mine = (threadIdx.x >> 6) * 64 + lane_;
        bf16x8 first[kFeedAhead];
#pragma unroll
        for (int i = 0; i < kFeedAhead; ++i) first[i] = ws[(size_t)i * kUnitVecs + mine];
#pragma unroll
        for (int i = 0; i < kFeedStage; ++i) stage[i] = ws[(size_t)(kFeedAhead + i) * kUnitVecs + mine];
#pragma unroll
        for (int i = 0; i < kFeedAhead; ++i) lds[i * kUnitVecs + mine] = first[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kFeedRead; ++i) read_unit_rt(frag[i], i);
        __syncthreads();
    }
    template <int PH>
    __device__ __forceinline__ const WUnit &cur() const { return frag[PH % 4]; }
    static constexpr int kLook = kFeedAhead + kFeedStage;
    template <class First>
    __device__ __forceinline__ void start_at(First &&first, unsigned lane_, bf16x8 *ring, const bf16x8 *)
    {
        lane = lane_;
        lds = ring;
        mine = (threadIdx.x >> 6) * 64 + lane_;
        bf16x8 head[kFeedAhead];
        static_for<0, kFeedAhead>([&](auto I) { head[decltype(I)::value] = first(I)[mine]; });
        static_for<0, kFeedStage>([&](auto I) {
            stage[decltype(I)::value] = first(std::integral_constant<int, kFeedAhead + decltype(I)::value>{})[mine];
        });
#pragma unroll
        for (int i = 0; i < kFeedAhead; ++i) lds[i * kUnitVecs + mine] = head[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kFeedRead; ++i) read_unit_rt(frag[i], i);
        __syncthreads();
    }
    template <int PH>
    __device__ __forceinline__ void advance_at(const bf16x8 *__restrict__ next)
    {
        // (FPN_FEED_NO*: timing-only builds that drop one part of the feed; wrong results)
#ifndef FPN_FEED_NOREAD
        read_unit<(PH + kFeedRead) % kFeedSlots>(frag[(PH + kFeedRead) % 4]);
#else
        opaque(frag[(PH + kFeedRead) % 4]);
#endif
#ifndef FPN_FEED_NOFILL
        lds[((PH + kFeedAhead) % kFeedSlots) * kUnitVecs + mine] = stage[PH % kFeedStage];
        stage[PH % kFeedStage] = next[mine];
#endif
    }
    template <int PH>
    __device__ __forceinline__ void advance(const bf16x8 *__restrict__ unit)
    {
        advance_at<PH>(unit + (size_t)(kFeedAhead + kFeedStage) * kUnitVecs);
    }
    template <int PH>
    __device__ __forceinline__ void sync()
    {
#ifndef FPN_FEED_NOBAR
        if constexpr (PH % 2 == 1) __syncthreads();
#endif
    }
};

// LdsFeed: the kernel's whole stream lies in LDS (a persistent workgroup copied it there
// once), followed by copies of its first kLdsAhead units, so that a wave which walks the
// stream again and again - once per group of points - finds the first units of its next
// round in its registers when it ends a round.  No memory reads, no barriers: unit x is read
// from LDS during step x - kLdsAhead.  The stream must hold an even number of units.
constexpr int kLdsAhead = 2;

struct LdsFeed {
    static constexpr int kMode = 2, kDepth = kRing;
    WUnit frag[kLdsAhead];
    const bf16x8 *lds;      // the stream in LDS
    const bf16x8 *base;     // the stream in memory: the layers name their units by address there
    unsigned lane;
    __device__ __forceinline__ void read_unit(WUnit &u, const bf16x8 *src) const
    {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int m = 0; m < 2; ++m) u.w[a][m] = src[(a * 2 + m) * 64 + lane];
    }
    __device__ __forceinline__ void start(const bf16x8 *__restrict__ ws, unsigned lane_, bf16x8 *copy)
    {
        lane = lane_;
        lds = copy;
        base = ws;
#pragma unroll
        for (int i = 0; i < kLdsAhead; ++i) read_unit(frag[i], lds + i * kUnitVecs);
    }
    template <int PH>
    __device__ __forceinline__ const WUnit &cur() const { return frag[PH % kLdsAhead]; }
    template <int PH>
    __device__ __forceinline__ void advance(const bf16x8 *__restrict__ unit)
    {
        read_unit(frag[PH % kLdsAhead], lds + (unit - base) + kLdsAhead * kUnitVecs);
    }
    template <int PH>
    __device__ __forceinline__ void sync() {}
};

// the workgroup's unit ring in LDS (CoopFeed), none for RingFeed
template <class Feed>
__device__ __forceinline__ bf16x8 *feed_ring()
{
    if constexpr (Feed::kMode == 1) {
        __shared__ bf16x8 wring[kFeedSlots * kUnitVecs];
        return wring;
    } else {
        return nullptr;
    }
}

// (timing-only build, wrong results: every layer constant is read from LDS instead of memory - what the
// kernels would gain if no constant load stood in the vector-memory queue between the weight loads)
#ifdef FPN_CONST_LDS_TIMING
__device__ __forceinline__ const float *const_src(const float *)
{
    __shared__ float dummy[2048];
    dummy[(threadIdx.x * 8u + blockIdx.x) % 2048u] = 0.001f * threadIdx.x;      // (keeps the reads from folding to nothing)
    return dummy;
}
#else
__device__ __forceinline__ const float *const_src(const float *p) { return p; }
#endif

// per-channel constants of a lane for one 32-channel tile: channels ch0 + 8q + i
struct TileConst {
    float4 q[4];
};

__device__ __forceinline__ float comp(const float4 &v, int i)
{
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

__device__ __forceinline__ void load_const(TileConst &c, const float *__restrict__ p, int ch0)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) c.q[q] = *reinterpret_cast<const float4 *>(p + ch0 + 8 * q);
}

// Layer constants of a wave, staged once in the wave's OWN piece of LDS (each wave reads only what it wrote:
// no barrier).  Read from memory where a layer wants them, a constant stands in the wave's vector-memory queue
// behind the weight units requested last - loads return in order, so waiting for it means waiting for the
// whole read-ahead of the weight ring; LDS reads are counted apart from the weight loads.  N floats (a
// multiple of 4), src 16-byte aligned.
// In two steps, for a kernel's opening: request() only asks for the vectors, put() writes them to LDS.
// A write placed right behind its own load makes the wave wait there - for every load asked for before it, the
// weight ring's first units included - and only then ask for its inputs: two trips to memory one after the
// other.  With every load of the opening requested first, the waits that follow are for one trip.
template <int N>
struct StagedConsts {
    static_assert(N % 4 == 0 && N <= 1024, "whole 16-byte vectors, four a lane at most");
    static constexpr int kVecs = (N / 4 + 63) / 64;
    typedef float vec4 __attribute__((ext_vector_type(4)));
    vec4 v0, v1, v2, v3;        // (register vectors by name: an array of float4 the compiler kept in scratch memory)
    template <int I>
    static __device__ __forceinline__ void get(vec4 &v, const float *__restrict__ src, unsigned lane)
    {
        const int k = I * 64 + (int)lane;
        if constexpr (I < kVecs) {
            if ((N / 4) % 64 == 0 || k < N / 4) v = reinterpret_cast<const vec4 *>(src)[k];
        }
    }
    template <int I>
    static __device__ __forceinline__ void set(float *dst, const vec4 &v, unsigned lane)
    {
        const int k = I * 64 + (int)lane;
        if constexpr (I < kVecs) {
            if ((N / 4) % 64 == 0 || k < N / 4) reinterpret_cast<vec4 *>(dst)[k] = v;
        }
    }
    __device__ __forceinline__ void request(const float *__restrict__ src, unsigned lane)
    {
        get<0>(v0, src, lane);
        get<1>(v1, src, lane);
        get<2>(v2, src, lane);
        get<3>(v3, src, lane);
    }
    __device__ __forceinline__ void put(float *dst, unsigned lane) const
    {
        set<0>(dst, v0, lane);
        set<1>(dst, v1, lane);
        set<2>(dst, v2, lane);
        set<3>(dst, v3, lane);
    }
};

// relu(acc + shift) as the next layer's operand, three instructions per register pair:
// packed add, packed conversion, and the clamp as a packed 16-bit INTEGER max with 0 on
// the bf16 bit patterns (negative floats, -0 included, are the negative integers).
__device__ __forceinline__ void finish(Act &o, const floatx16 &a, const TileConst &sh)
{
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        shortx8 bits;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const int r = 8 * s + e;
#ifdef FPN_NOADD_TIMING     // (timing-only build, wrong results: what the epilogues' shift additions cost)
            const floatx2 y = {a[r], a[r + 1]};
#else
            const floatx2 y = {a[r] + comp(sh.q[r >> 2], r & 3), a[r + 1] + comp(sh.q[r >> 2], (r & 3) + 1)};
#endif
            const shortx2 v = __builtin_elementwise_max(__builtin_bit_cast(shortx2, __builtin_convertvector(y, bf16x2)),
                                                        shortx2{0, 0});
            bits[e] = v[0];
            bits[e + 1] = v[1];
        }
        o.m[s] = __builtin_bit_cast(bf16x8, bits);
    }
}

// ... or kept in fp32 (the layer before the final 128 -> 2 product)
__device__ __forceinline__ void finish(floatx16 &o, const floatx16 &a, const TileConst &sh)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = fmaxf(a[r] + comp(sh.q[r >> 2], r & 3), 0.f);
}

__device__ __forceinline__ floatx16 mfma(const bf16x8 &a, const bf16x8 &b, const floatx16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// The shift can also enter as the accumulator's starting value (the C operand of a tile's
// first MFMA: register r of every lane is channel (r & 3) + 8 (r >> 2) + 4 h, what a
// TileConst holds) - then the epilogue is conversion and clamp alone, 16 instructions per
// tile instead of 24.  The shifts must be there when the tile's first MFMA issues.
__device__ __forceinline__ floatx16 as_acc(const TileConst &c)
{
    floatx16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = comp(c.q[r >> 2], r & 3);
    return v;
}

__device__ __forceinline__ void finish0(Act &o, const floatx16 &a)
{
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        shortx8 bits;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const int r = 8 * s + e;
            const shortx2 v = __builtin_elementwise_max(
                __builtin_bit_cast(shortx2, __builtin_convertvector(floatx2{a[r], a[r + 1]}, bf16x2)), shortx2{0, 0});
            bits[e] = v[0];
            bits[e + 1] = v[1];
        }
        o.m[s] = __builtin_bit_cast(bf16x8, bits);
    }
}

// One unit: input tile J of each of the wave's NP point groups against the two output
// tiles of the pair (4 NP MFMAs, every weight fragment used NP times).  FIRST: the
// accumulators start from zero (the hardware takes the constant, no register clearing).
// T = true: transposed product (weights are the A operand, result lane = point);
// T = false: lane = output channel, registers = points.
// NTILE = 1: the pair's second tile is padding (an odd number of output tiles): its
// MFMAs are left out.
template <bool FIRST, bool T, int J, int NTILE = 2, int NP, int NJ>
__device__ __forceinline__ void unit_mfma(floatx16 (&acc)[2][NP], const WUnit &u, const Act (&in)[NP][NJ])
{
    const floatx16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < NP; ++g)
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                const floatx16 c = (FIRST && m == 0) ? z : acc[t][g];
                acc[t][g] = T ? mfma(u.w[t][m], in[g][J].m[m], c) : mfma(in[g][J].m[m], u.w[t][m], c);
            }
}

__device__ __forceinline__ void finish0(floatx16 &o, const floatx16 &a)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = fmaxf(a[r], 0.f);
}

// ... the same for the first input tile of a pair, the accumulators starting from the shifts
template <int NTILE, int NP, int NJ>
__device__ __forceinline__ void unit_mfma_from(floatx16 (&acc)[2][NP], const WUnit &u, const Act (&in)[NP][NJ],
                                               const TileConst (&init)[2])
{
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < NP; ++g)
#pragma unroll
            for (int t = 0; t < NTILE; ++t)
                acc[t][g] = mfma(u.w[t][m], in[g][0].m[m], m == 0 ? as_acc(init[t]) : acc[t][g]);
}

// Instruction order inside a unit: the wave issues in order, and a vector-ALU or memory
// instruction placed after the unit's last MFMA would run with the matrix pipe idle.  So
// after each MFMA come one memory read (while there are any, NVMEM of them) and up to
// NVALU vector-ALU instructions - they issue in the 32 cycles the MFMA occupies the pipe;
// nothing crosses the end of the unit.
// MODE 1 (CoopFeed): the unit's four weight loads are one load, four LDS reads and one LDS
// write instead, all placed behind the unit's FIRST MFMA: the barrier that may end the
// unit waits for every LDS operation of the wave, so they get the rest of the unit's MFMAs
// to complete in.
template <int MODE, int NMFMA, int NVMEM, int NVALU, int NHOOK = 0>
__device__ __forceinline__ void sched_unit()
{
    constexpr bool COOP = MODE == 1;
    if constexpr (MODE == 2) {      // LdsFeed: the weights and the layer constants are all LDS reads;
        // NHOOK: memory reads a caller's hook issued in this unit (the next round's inputs), one behind each MFMA
        constexpr int PER = (NVMEM + NMFMA - 1) / NMFMA, HPER = (NHOOK + NMFMA - 1) / NMFMA;
        static_for<0, NMFMA>([&](auto M) {
            constexpr int m = decltype(M)::value;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (m * PER < NVMEM) __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
            if constexpr (m * HPER < NHOOK) {       // (with the address arithmetic of the reads)
                __builtin_amdgcn_sched_group_barrier(0x020, HPER, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NVALU + 4 * HPER, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    constexpr int NV = COOP ? NVMEM - 3 : NVMEM;
#pragma unroll
    for (int m = 0; m < NMFMA; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (COOP && m == 0) {
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        if (m < NV) __builtin_amdgcn_sched_group_barrier(0x020, (NV + NMFMA - 1) / NMFMA, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Transposed layer: out[g][NT] = relu(W'^T . in[g][NJ] + shift), W' = W . diag(scale).
// ws: the layer's first unit; the ring holds units [0, kRing) of it on entry (slot
// (R0 + i) % kRing = unit i) and the first kRing units after the layer on exit.  The
// epilogue of a pair rides on the first two units of the next pair; the last one is
// exposed.  Out: Act (bf16 operand) or floatx16 (fp32).
template <int NP, int NJ, int NT, int R0, class Out, class Feed>
__device__ __forceinline__ void layer_t(Out (&out)[NP][NT], const Act (&in)[NP][NJ], const bf16x8 *__restrict__ ws,
                                        const float *__restrict__ sh, Feed &feed, int h)
{
    // an odd NT: the stream still holds whole pairs (the layer's columns zero-padded to a
    // multiple of 64), the last pair's second tile is neither multiplied nor finished
    static_assert(NJ >= 2, "unsupported layer shape");
    constexpr int NPAIR = (NT + 1) / 2;
    constexpr int U = NPAIR * NJ;
    floatx16 acc[2][2][NP];
    TileConst csh[2];
    sh = const_src(sh);
    static_for<0, U>([&](auto I) {
        constexpr int S = decltype(I)::value, tp = S / NJ, j = S % NJ, set = tp & 1;
        constexpr int slot = (R0 + S) % Feed::kDepth;
        constexpr int ntile = 2 * tp + 1 < NT ? 2 : 1;
        unit_mfma<j == 0, true, j, ntile>(acc[set], feed.template cur<slot>(), in);
        feed.template advance<slot>(ws + (size_t)S * kUnitVecs);
        if constexpr (tp > 0 && j < 2) {
#pragma unroll
            for (int g = 0; g < NP; ++g) finish(out[g][2 * (tp - 1) + j], acc[set ^ 1][j][g], csh[j]);
        }
        if constexpr (j == NJ - 1) {          // shifts of this pair, wanted one unit from now
            load_const(csh[0], sh, 32 * (2 * tp) + 4 * h);
            if constexpr (ntile == 2) load_const(csh[1], sh, 32 * (2 * tp + 1) + 4 * h);
        }
        sched_unit<Feed::kMode, 2 * ntile * NP, j == NJ - 1 ? 4 + 4 * ntile : 4,
                   (tp > 0 && j < 2) ? (48 * NP + 2 * ntile * NP - 1) / (2 * ntile * NP) : 1>();
        feed.template sync<slot>();
    });
    constexpr int last = (NPAIR - 1) & 1;
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        finish(out[g][2 * NPAIR - 2], acc[last][0][g], csh[0]);
        if constexpr (NT % 2 == 0) finish(out[g][NT - 1], acc[last][1][g], csh[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// The transposed layer with the shifts as the accumulators' starting values: the epilogue
// of a tile is conversion and clamp alone, 16 vector instructions instead of 24 (one wave per
// SIMD issues in order: with five vector instructions per MFMA the stream itself, not the
// matrix pipe, set the pace of the v1 kernels).  For shifts that lie in LDS - a pair's are
// read during the last unit of the pair before; with shifts in memory the same arithmetic,
// but the read then waits behind the weight ring's loads.  csh: the shifts of the layer's
// FIRST pair, read by the caller early enough.  relu(acc + shift) with the sum started at
// 0 and the sum started at the shift round differently: kernels that are compared bit for
// bit use the same form.
struct NoHook {
    template <class I>
    __device__ __forceinline__ void operator()(I) const {}
};

// hook(integral_constant<unit>): called once in every unit, for work of the caller's that should issue between
// this layer's MFMAs (NHOOK memory reads per unit at most)
template <int NP, int NJ, int NT, int R0, int NHOOK = 0, class Out, class Feed, class Hook = NoHook>
__device__ __forceinline__ void layer_t_from(Out (&out)[NP][NT], const Act (&in)[NP][NJ], const bf16x8 *__restrict__ ws,
                                             const float *__restrict__ sh, Feed &feed, int h, TileConst (&csh)[2],
                                             Hook &&hook = Hook())
{
    static_assert(NJ >= 2, "unsupported layer shape");
    constexpr int NPAIR = (NT + 1) / 2;
    constexpr int U = NPAIR * NJ;
    floatx16 acc[2][2][NP];
    static_for<0, U>([&](auto I) {
        constexpr int S = decltype(I)::value, tp = S / NJ, j = S % NJ, set = tp & 1;
        constexpr int slot = (R0 + S) % Feed::kDepth;
        constexpr int ntile = 2 * tp + 1 < NT ? 2 : 1;
        if constexpr (j == 0) unit_mfma_from<ntile>(acc[set], feed.template cur<slot>(), in, csh);
        else unit_mfma<false, true, j, ntile>(acc[set], feed.template cur<slot>(), in);
        feed.template advance<slot>(ws + (size_t)S * kUnitVecs);
        if constexpr (tp > 0 && j < 2) {
#pragma unroll
            for (int g = 0; g < NP; ++g) finish0(out[g][2 * (tp - 1) + j], acc[set ^ 1][j][g]);
        }
        constexpr bool more = j == NJ - 1 && tp + 1 < NPAIR;
        constexpr int nnext = 2 * (tp + 1) + 1 < NT ? 2 : 1;
        if constexpr (more) {
            load_const(csh[0], sh, 32 * (2 * tp + 2) + 4 * h);
            if constexpr (nnext == 2) load_const(csh[1], sh, 32 * (2 * tp + 3) + 4 * h);
        }
        hook(I);
        sched_unit<Feed::kMode, 2 * ntile * NP, more ? 4 + 4 * nnext : 4,
                   (tp > 0 && j < 2) ? (32 * NP + 2 * ntile * NP - 1) / (2 * ntile * NP) : 1, NHOOK>();
        feed.template sync<slot>();
    });
    constexpr int last = (NPAIR - 1) & 1;
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        finish0(out[g][2 * NPAIR - 2], acc[last][0][g]);
        if constexpr (NT % 2 == 0) finish0(out[g][NT - 1], acc[last][1][g]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// layer_t_from with the first pair's shifts read here (constants that lie in the wave's LDS: the read is
// short and stands in no queue behind the weight loads)
template <int NP, int NJ, int NT, int R0, class Out, class Feed>
__device__ __forceinline__ void layer_t_acc(Out (&out)[NP][NT], const Act (&in)[NP][NJ], const bf16x8 *__restrict__ ws,
                                            const float *__restrict__ sh, Feed &feed, int h)
{
    TileConst csh[2];
    load_const(csh[0], sh, 4 * h);
    if constexpr (NT > 1) load_const(csh[1], sh, 32 + 4 * h);
    layer_t_from<NP, NJ, NT, R0>(out, in, ws, sh, feed, h, csh);
}

// max over the wave's points of relu(acc + shift) for the lane's channel, merged into
// dst[channel] with an integer atomic max (values >= +0 order like their bit patterns).
// dst: the output row in memory, or the workgroup's LDS copy of it (see WgPool).
template <int NP>
__device__ __forceinline__ void max_points(const floatx16 (&acc)[NP], float sh, int *dst)
{
    // two chains of three-operand maxima (v_max3_f32: one instruction per two values)
    float m0 = acc[0][0], m1 = acc[0][1];
#pragma unroll
    for (int g = 0; g < NP; ++g)
#pragma unroll
        for (int r = (g == 0 ? 2 : 0); r < 16; r += 4) {
            m0 = fmaxf(fmaxf(m0, acc[g][r]), acc[g][r + 1]);
            if (r + 2 < 16) m1 = fmaxf(fmaxf(m1, acc[g][r + 2]), acc[g][r + 3]);
        }
    atomicMax(dst, __float_as_int(fmaxf(fmaxf(m0, m1) + sh, 0.f)));
}

// Maxima over points leave a wave as one atomic per (32-channel tile, lane); at bf16
// speed the memory-side atomic units, not the matrix cores, would set the pace (measured:
// the seg head takes 0.82 ms with, 0.42 ms without them).  So the four waves of a
// workgroup first merge their maxima in LDS (ds_max, no return value, nothing waits on
// it) and the workgroup then issues one memory atomic per channel.  That needs every wave
// of the workgroup to have rows and the pooling groups (g rows each: a frustum's points,
// a ball's samples) to tile the workgroup's rows evenly; otherwise the waves update
// memory directly.
struct WgPool {
    bool on;          // workgroup-uniform
    int groups;       // pooling groups inside this workgroup's rows (1, 2 or 4)
    int local;        // this wave's group among them
    long first;       // index of the workgroup's first group
};

// Flipout: each frustum copy has its own weight stream (fpn_flipout_stream_bf16).  Its
// workgroups are then placed on ONE XCD (consecutive workgroup ids go round the eight
// XCDs, so id i handles logical block (i % 8) * (grid / 8) + i / 8) and read the copy out
// of that XCD's L2.
__device__ __forceinline__ unsigned logical_block(long per_frustum)
{
    const unsigned g = gridDim.x, i = blockIdx.x;
    return (per_frustum && g % 8 == 0) ? (i % 8) * (g / 8) + i / 8 : i;
}

// a wave-uniform index as a scalar (the wave number comes from threadIdx)
__device__ __forceinline__ size_t scalar_index(long v)
{
    return (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

template <int NP>
__device__ __forceinline__ WgPool wg_pool(long rows, int g, unsigned block)
{
    constexpr int kRows = kWaves * 32 * NP;
    const long r0 = (long)block * kRows;
    WgPool w;
    w.on = r0 + kRows <= rows && (g % kRows == 0 || kRows % g == 0);
    w.groups = g >= kRows ? 1 : kRows / g;
    w.local = g >= kRows ? 0 : (int)((threadIdx.x >> 6) * (32 * NP) / g);
    w.first = r0 / g;
    return w;
}

template <int C>
__device__ __forceinline__ void wg_pool_clear(int (&lmax)[4 * C], const WgPool &w)
{
    for (int i = threadIdx.x; i < w.groups * C; i += 64 * kWaves) lmax[i] = 0;
    __syncthreads();
}

// after the last layer: LDS maxima -> memory; out row of group k at out + (first + k) * ld
template <int C>
__device__ __forceinline__ void wg_pool_flush(int (&lmax)[4 * C], const WgPool &w, float *out, int ld, int cvalid)
{
    __syncthreads();
    for (int k = 0; k < w.groups; ++k)
        for (int ch = threadIdx.x; ch < cvalid; ch += 64 * kWaves)
            atomicMax(reinterpret_cast<int *>(out + (w.first + k) * ld + ch), lmax[k * C + ch]);
}

// Last layer of a stack: gmax[0 : 32 NT] = max(gmax, max over the wave's points of
// relu(in . W' + shift)), four output tiles per trip of a rolled loop; the reduction of
// one pair rides on the units of the next.  Channels >= cvalid (widths padded up to a
// multiple of 128) are not written.
template <int NP, int NJ, int NT, int R0, class Feed>
__device__ __forceinline__ void layer_max(const Act (&in)[NP][NJ], const bf16x8 *__restrict__ ws,
                                          const float *__restrict__ sh, int *gmax, Feed &feed, unsigned lane,
                                          int cvalid = 32 * NT)
{
    // (a trip of the loop below takes 2 NJ units: with more than one trip the ring and feed
    // phases must come round)
    static_assert(NJ >= 2 && NT % 4 == 0 && (NT == 4 || (2 * NJ) % kRing == 0), "unsupported layer shape");
    const int c = lane & 31;
    floatx16 acc[2][2][NP];
    const floatx16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        acc[1][0][g] = z;
        acc[1][1][g] = z;
    }
    float rsh[2] = {0.f, 0.f};
    sh = const_src(sh);
#pragma unroll 1
    for (int t0 = 0; t0 < NT; t0 += 4) {
        const bf16x8 *wt = ws + (size_t)(t0 / 2) * NJ * kUnitVecs;
        static_for<0, 2 * NJ>([&](auto I) {
            constexpr int S = decltype(I)::value, set = S / NJ, j = S % NJ;
            constexpr int slot = (R0 + S) % Feed::kDepth;
            unit_mfma<j == 0, false, j>(acc[set], feed.template cur<slot>(), in);
            feed.template advance<slot>(wt + (size_t)S * kUnitVecs);
            // the other accumulator pair holds the two tiles before these; the very first
            // time there are none (it holds zeros, and so does the shift): the update is
            // then a harmless max with 0 - no branch inside the unit
            const int tcand = set == 0 ? t0 - 2 : t0;
            const bool has_prev = (set == 1 || t0 > 0) && 32 * tcand < cvalid;
            const int tprev = has_prev ? tcand : 0;
            if constexpr (j < 2) max_points(acc[set ^ 1][j], has_prev ? rsh[j] : 0.f, gmax + 32 * (tprev + j) + c);
            if constexpr (j == NJ - 1) {
                const int tcur = t0 + 2 * set;
                rsh[0] = sh[32 * tcur + c];
                rsh[1] = sh[32 * (tcur + 1) + c];
            }
            sched_unit<Feed::kMode, 4 * NP, j == NJ - 1 ? 6 : 4, j < 2 ? (12 * NP + 8 + 4 * NP - 1) / (4 * NP) : 1>();
            feed.template sync<slot>();
        });
    }
    if (32 * (NT - 2) < cvalid) {
        max_points(acc[1][0], rsh[0], gmax + 32 * (NT - 2) + c);
        max_points(acc[1][1], rsh[1], gmax + 32 * (NT - 1) + c);
    }
}

// First layer of a stack (K = 3..8 inputs straight from memory: coordinates in metres, which
// one bf16 number would quantise to decimetres): out[NT] = relu(scale * (W1^T . x^T) + shift)
// as bf16 operands, on the bf16 matrix instruction with SPLIT operands.  Every input and
// every (scaled) weight is written as the sum of two bf16 numbers, x = xh + xl, w = wh + wl
// (what is left over is below 2^-16 of the value), and the product taken as
// xh wh + xl wh + xh wl, exact in the fp32 accumulator: three slots of the instruction's
// 16-deep contraction per input channel.  KP <= 5 channels fit one MFMA per output tile
// and 32 points (32 clocks), KP = 8 take two; the fp32 instruction (two channels per
// 64-clock MFMA, then a multiply-add per result) cost a quarter of a set-abstraction round.
// Slot q = 16 f + 8 h + e (MFMA f, lane half h, element e) holds channel q % KP, term
// q / KP (0: xh wh, 1: xl wh, 2: xh wl, above: nothing); two more slots hold the shift, split
// the same way, against inputs of 1 - no constant to add afterwards.  x: all KP channels of the lane's
// point (both lane halves hold the same point; zero beyond the channel count); w1
// (K, 32 NT) row-major fp32.
//
// The layer's weights and constants are the same for every point group of the wave and
// depend on nothing the kernel computes: FirstLayer::load is issued BEFORE the kernel
// gathers its inputs, so that the two memory latencies overlap.
template <int NT, int KP>
struct FirstLayer {
    static constexpr int NMF = (3 * KP + 2 + 15) / 16;
    static_assert(KP >= 3 && KP <= 8 && NMF <= 2, "unsupported channel count");
    bf16x8 wa[NT][NMF];

    // the lane's fragment f: element e is slot 16 f + 8 h + e, picked from the channels' high and
    // low parts; slots 3 KP and 3 KP + 1 carry the shift (one0 = its high part, one1 = its low
    // part) against inputs of 1
    __device__ static __forceinline__ bf16x8 fragment(const __bf16 (&hi)[KP], const __bf16 (&lo)[KP], __bf16 one0, __bf16 one1,
                                                      int f, int h, bool weights)
    {
        bf16x8 v[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int q = 16 * f + 8 * hh + e, ch = q % KP, term = q / KP;
                // weights: wh, wh, wl; inputs: xh, xl, xh
                v[hh][e] = q == 3 * KP ? one0 : q == 3 * KP + 1 ? one1 : term > 2 ? (__bf16)0.f
                                                                 : (term == (weights ? 2 : 1) ? lo[ch] : hi[ch]);
            }
        const int4 a = __builtin_bit_cast(int4, v[0]), b = __builtin_bit_cast(int4, v[1]);
        return __builtin_bit_cast(bf16x8, make_int4(h ? b.x : a.x, h ? b.y : a.y, h ? b.z : a.z, h ? b.w : a.w));
    }

    __device__ __forceinline__ void load(const float *__restrict__ w1, int k_total, const float *__restrict__ sc,
                                         const float *__restrict__ sh, unsigned lane, int h)
    {
        const int c = lane & 31;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float scale = sc[32 * t + c], shift = sh[32 * t + c];
            __bf16 hi[KP], lo[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                // (the load itself is unconditional - row 0 stands in past the last channel: a load inside a
                // branch makes the wave wait there for every load requested so far, a trip to memory per channel)
                const float wk = w1[(k < k_total ? k : 0) * (32 * NT) + 32 * t + c];
                const float w = k < k_total ? wk * scale : 0.f;
                hi[k] = (__bf16)w;
                lo[k] = (__bf16)(w - (float)hi[k]);
            }
            const __bf16 s0 = (__bf16)shift, s1 = (__bf16)(shift - (float)s0);
#pragma unroll
            for (int f = 0; f < NMF; ++f) wa[t][f] = fragment(hi, lo, s0, s1, f, h, true);
        }
    }

    // all point groups of the wave: the MFMAs first (they are independent), then the epilogues.
    // init: NULL, or per group and tile what other inputs of the layer have contributed already
    // (scaled like the weights; register r of a lane = channel 8 (r >> 2) + 4 h + (r & 3) of the tile).
    template <int NP>
    __device__ __forceinline__ void apply(Act (&out)[NP][NT], const float (&x)[NP][KP], int h,
                                          const floatx16 (*init)[NT] = nullptr) const
    {
        const floatx16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        floatx16 acc[NP][NT];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            __bf16 hi[KP], lo[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                hi[k] = (__bf16)x[g][k];
                lo[k] = (__bf16)(x[g][k] - (float)hi[k]);
            }
            bf16x8 xb[NMF];
#pragma unroll
            for (int f = 0; f < NMF; ++f) xb[f] = fragment(hi, lo, (__bf16)1.f, (__bf16)1.f, f, h, false);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[g][t] = mfma(wa[t][0], xb[0], init ? init[g][t] : z);
                if constexpr (NMF == 2) acc[g][t] = mfma(wa[t][1], xb[1], acc[g][t]);
            }
        }
#pragma unroll
        for (int g = 0; g < NP; ++g)
#pragma unroll
            for (int t = 0; t < NT; ++t) finish0(out[g][t], acc[g][t]);
        __builtin_amdgcn_sched_barrier(0);
    }
};

constexpr int units_of(int cin, int cout) { return (cout / 64) * (cin / 32); }

// ---------------------------------------------------------------------------
// seg head: conv1 .. conv5 + max over points.  Stream: conv2, conv3, conv4, conv5.
// point_feat (conv3's output) goes to memory as the tail's operand fragments: per point
// 64 bf16, fragment (tile t, s) of lane half h at element ((2t + s) * 2 + h) * 8.
// A wave takes 32 NP consecutive points of one frustum.
// ---------------------------------------------------------------------------
struct HeadParams {
    long rows;
    int n, cpc;
    const float *pc;
    const float *w1, *sc1, *sh1;
    const bf16x8 *ws;
    const float *sh2, *sh3, *sh4, *sh5;
    bf16x8 *point_feat;
    float *gmax;
    int ldg;
    long copy_vecs;     // > 0: one stream (and one w1) per frustum, this many 16-byte vectors apart
};

template <int NP, class Feed = RingFeed>
__global__ __launch_bounds__(64 * kWaves) void seg_head_bf16_kernel(const HeadParams pin)
{
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const unsigned block = logical_block(pin.copy_vecs);
    const long row0 = ((long)block * kWaves + (threadIdx.x >> 6)) * (32 * NP);
    __shared__ int lmax[4 * 1024];
    const WgPool pool = wg_pool<NP>(pin.rows, pin.n, block);
    if (!pool.on && row0 >= pin.rows) return;
    HeadParams p = pin;
    if (pin.copy_vecs) {
        const size_t f = scalar_index(row0 / pin.n);
        p.ws += f * pin.copy_vecs;
        p.w1 += f * (size_t)(pin.cpc * 64);
    }
    if (pool.on) wg_pool_clear<1024>(lmax, pool);
    int *gmax = reinterpret_cast<int *>(p.gmax + (row0 / p.n) * p.ldg);

#ifdef FPN_HEAD_STAMPS      // timing-only build: cycle stamps of one wave into the frustum's maxima row (wrong results)
    long long stamp[8];
    stamp[0] = clock64();
#define FPN_HSTAMP(i) stamp[i] = clock64()
#else
#define FPN_HSTAMP(i)
#endif
    Feed feed;
    feed.start(p.ws, lane, feed_ring<Feed>());
    __shared__ float hconst[kWaves][64 + 64 + 128 + 1024];
    float *lc = hconst[threadIdx.x >> 6];
    StagedConsts<64> c2, c3;
    StagedConsts<128> c4;
    StagedConsts<1024> c5;
    c2.request(p.sh2, lane);
    c3.request(p.sh3, lane);
    c4.request(p.sh4, lane);
    c5.request(p.sh5, lane);
    const float *sh2 = lc, *sh3 = lc + 64, *sh4 = lc + 128, *sh5 = lc + 256;
    Act a1[NP][2], a2[NP][2], a3[NP][2], a4[NP][4];
    {
        FirstLayer<2, 8> first;
        first.load(p.w1, p.cpc, p.sc1, p.sh1, lane, h);
        float x[NP][8];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const float *src = p.pc + (row0 + 32 * g + pt) * p.cpc;
#pragma unroll
            for (int k = 0; k < 8; ++k) {       // (unconditional loads, as in FirstLayer::load)
                const float xk = src[k < p.cpc ? k : 0];
                x[g][k] = k < p.cpc ? xk : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // (the constants' writes stay behind the requests above)
        c2.put(lc, lane);
        c3.put(lc + 64, lane);
        c4.put(lc + 128, lane);
        c5.put(lc + 256, lane);
        FPN_HSTAMP(1);
        first.apply(a1, x, h);
    }
    FPN_HSTAMP(2);
    constexpr int U2 = units_of(64, 64), U3 = units_of(64, 64), U4 = units_of(64, 128);
    const bf16x8 *ws = p.ws;
    layer_t_acc<NP, 2, 2, 0>(a2, a1, ws, sh2, feed, h);
    FPN_HSTAMP(3);
    ws += U2 * kUnitVecs;
    layer_t_acc<NP, 2, 2, U2 % kRing>(a3, a2, ws, sh3, feed, h);
    FPN_HSTAMP(4);
    ws += U3 * kUnitVecs;
#pragma unroll
    for (int g = 0; g < NP; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) p.point_feat[(row0 + 32 * g + pt) * 8 + (2 * t + s) * 2 + h] = a3[g][t].m[s];
    layer_t_acc<NP, 2, 4, (U2 + U3) % kRing>(a4, a3, ws, sh4, feed, h);
    FPN_HSTAMP(5);
    ws += U4 * kUnitVecs;
    if (pool.on) {      // (two copies of the layer: LDS and memory atomics are different instructions)
        layer_max<NP, 4, 32, (U2 + U3 + U4) % kRing>(a4, ws, sh5, lmax + pool.local * 1024, feed, lane);
        FPN_HSTAMP(6);
        wg_pool_flush<1024>(lmax, pool, p.gmax, p.ldg, 1024);
#ifdef FPN_HEAD_STAMPS
        stamp[7] = clock64();
        if (blockIdx.x == gridDim.x / 2 + 8 && threadIdx.x == 64)
            for (int i = 0; i < 8; ++i) reinterpret_cast<long long *>(p.gmax)[32 + i] = stamp[i];
#endif
    } else {
        layer_max<NP, 4, 32, (U2 + U3 + U4) % kRing>(a4, ws, sh5, gmax, feed, lane);
    }
}

// ---------------------------------------------------------------------------
// seg head, a WORKGROUP PER FRUSTUM.  seg_head_bf16_kernel's waves live for one round of 32 NP points: of a
// wave's 36 k cycles 6.6 k pass before its first MFMA (parameters, first-layer weights, the points and the
// first weight units all come from memory, and with one wave per SIMD nothing else runs meanwhile), and conv4
// waits 2.7 k for the point_feat stores in front of its weight loads (stores and loads complete in order).
// Here a workgroup takes a whole frustum: its points come into LDS in one sweep, the first layer's operands,
// the layer constants and the ring are set up once, and the waves then walk the frustum in rounds of
// kWaves x 32 NP points - the ring wraps (WrapRingFeedT), so a round starts with its first units loaded; a
// round's point_feat stores go out behind its last layer, where only the next round's first waits see them;
// the maxima of all rounds meet in LDS and leave with one atomic per channel.  Same arithmetic per point
// as seg_head_bf16_kernel: same bits.  Needs n % (kWaves * 32 NP) == 0 and n * cpc <= kHeadXin floats.
// ---------------------------------------------------------------------------
constexpr int kHeadXin = 8192;

template <int NP, class Feed>
__global__ __launch_bounds__(64 * kWaves) void seg_head_frustum_bf16_kernel(const HeadParams pin)
{
    constexpr int kRound = kWaves * 32 * NP;
    constexpr int U2 = units_of(64, 64), U3 = units_of(64, 64), U4 = units_of(64, 128), U5 = units_of(128, 1024);
    static_assert((U2 + U3 + U4 + U5) % kRing == 0, "the ring's slots must come round with the stream");
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    __shared__ int lmax[1024];
    __shared__ float hconst[64 + 64 + 128 + 1024];
    __shared__ float xin[kHeadXin];
    // a workgroup per frustum (grid = frustums), or fewer workgroups that each walk every gridDim-th frustum
    const long frustums = pin.rows / pin.n;
#pragma unroll 1
    for (long f = gridDim.x == frustums ? (long)logical_block(pin.copy_vecs) : (long)blockIdx.x; f < frustums; f += gridDim.x) {
    HeadParams p = pin;
    if (pin.copy_vecs) {
        p.ws += (size_t)f * pin.copy_vecs;
        p.w1 += (size_t)f * (size_t)(pin.cpc * 64);
    }

#ifdef FPN_HEAD_STAMPS
    long long stamp[12];
    stamp[0] = clock64();
#endif
    Feed feed;
    feed.start(p.ws, lane, feed_ring<Feed>());
    feed.end = p.ws + (size_t)(U2 + U3 + U4 + U5) * kUnitVecs;
    feed.len = (long)(U2 + U3 + U4 + U5) * kUnitVecs;
    FirstLayer<2, 8> first;
    first.load(p.w1, p.cpc, p.sc1, p.sh1, lane, h);
    {
        // every load of the set-up is requested before its first LDS write, and without a branch round any of
        // them (indices past the end read vector 0 instead; only the writes are conditional): a write right
        // behind its load, or a load inside a branch, makes the wave wait for all loads requested so far - a trip
        // to memory per step instead of one
        typedef float vec4 __attribute__((ext_vector_type(4)));
        constexpr int kThreads = 64 * kWaves, kXv = kHeadXin / 4 / kThreads;
        const int vecs = p.n * p.cpc / 4;
        const vec4 *src = reinterpret_cast<const vec4 *>(p.pc + (size_t)f * p.n * p.cpc);
        vec4 xv[kXv];
#pragma unroll
        for (int i = 0; i < kXv; ++i) {
            const int k = i * kThreads + (int)threadIdx.x;
            xv[i] = src[k < vecs ? k : 0];
        }
        // the layers' constants, 320 vectors in hconst's order: conv2's 16, conv3's 16, conv4's 32, conv5's 256
        constexpr int kConstVecs = (64 + 64 + 128 + 1024) / 4;
        static_assert(kConstVecs <= 2 * kThreads, "two vectors a thread");
        auto const_vec = [&](int k) {
            const float *base = k < 16 ? p.sh2 : k < 32 ? p.sh3 - 64 : k < 64 ? p.sh4 - 128 : p.sh5 - 256;
            return reinterpret_cast<const vec4 *>(base)[k];
        };
        const int k1 = (int)threadIdx.x + kThreads;
        const vec4 cv0 = const_vec((int)threadIdx.x), cv1 = const_vec(k1 < kConstVecs ? k1 : 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < kXv; ++i) {
            const int k = i * kThreads + (int)threadIdx.x;
            if (k < vecs) reinterpret_cast<vec4 *>(xin)[k] = xv[i];
        }
        for (int i = threadIdx.x; i < 1024; i += kThreads) lmax[i] = 0;
        reinterpret_cast<vec4 *>(hconst)[threadIdx.x] = cv0;
        if (k1 < kConstVecs) reinterpret_cast<vec4 *>(hconst)[k1] = cv1;
    }
    __syncthreads();
    const float *sh2 = hconst, *sh3 = hconst + 64, *sh4 = hconst + 128, *sh5 = hconst + 256;
    const bf16x8 *ws = p.ws;

#pragma unroll 1
    for (int r0 = (int)wave * (32 * NP); r0 < p.n; r0 += kRound) {
#ifdef FPN_HEAD_STAMPS
        const bool st = r0 / kRound == 1;
        if (r0 / kRound == 0) stamp[1] = clock64();
        if (st) stamp[2] = clock64();
#define FPN_FSTAMP(i) if (st) stamp[i] = clock64()
#else
#define FPN_FSTAMP(i)
#endif
        Act a1[NP][2], a2[NP][2], a3[NP][2], a4[NP][4];
        {
            float x[NP][8];
#pragma unroll
            for (int g = 0; g < NP; ++g) {
                const float *src = xin + (r0 + 32 * g + pt) * p.cpc;
#pragma unroll
                for (int k = 0; k < 8; ++k) {       // (unconditional reads, as in FirstLayer::load)
                    const float xk = src[k < p.cpc ? k : 0];
                    x[g][k] = k < p.cpc ? xk : 0.f;
                }
            }
            first.apply(a1, x, h);
        }
        FPN_FSTAMP(3);
        layer_t_acc<NP, 2, 2, 0>(a2, a1, ws, sh2, feed, h);
        FPN_FSTAMP(4);
        layer_t_acc<NP, 2, 2, U2 % kRing>(a3, a2, ws + U2 * kUnitVecs, sh3, feed, h);
        FPN_FSTAMP(5);
        const long row0 = f * p.n + r0;
        auto store_feat = [&]() {       // point_feat = conv3's output
#pragma unroll
            for (int g = 0; g < NP; ++g)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        p.point_feat[(row0 + 32 * g + pt) * 8 + (2 * t + s2) * 2 + h] = a3[g][t].m[s2];
        };
        layer_t_acc<NP, 2, 4, (U2 + U3) % kRing>(a4, a3, ws + (U2 + U3) * kUnitVecs, sh4, feed, h);
        FPN_FSTAMP(6);
        layer_max<NP, 4, 32, (U2 + U3 + U4) % kRing>(a4, ws + (U2 + U3 + U4) * kUnitVecs, sh5, lmax, feed, lane);
        FPN_FSTAMP(7);
        store_feat();     // (behind the last layer: only the next round's first waits stand behind these stores)
        FPN_FSTAMP(8);
    }
#ifdef FPN_HEAD_STAMPS
    stamp[9] = clock64();
#endif
    __syncthreads();
    for (int ch = threadIdx.x; ch < 1024; ch += 64 * kWaves)
        atomicMax(reinterpret_cast<int *>(p.gmax + f * p.ldg + ch), lmax[ch]);
#ifdef FPN_HEAD_STAMPS
    stamp[10] = clock64();
    if (blockIdx.x == gridDim.x / 2 + 8 && threadIdx.x == 64)
        for (int i = 0; i < 11; ++i) reinterpret_cast<long long *>(pin.gmax)[32 + i] = stamp[i];
#endif
    // (the next frustum's set-up writes xin and hconst, which every wave has finished reading at the barrier
    // above, and lmax entries that this same thread has just read)
    }
}

// ---------------------------------------------------------------------------
// seg tail: conv6 (point part; the frustum part arrives as a per-frustum shift) ..
// conv10.  Stream: conv6 (64 -> 512), conv7, conv8, conv9.  32 points per wave: conv6's
// 512-wide output (128 registers as bf16) leaves no room for a second point group.
// ---------------------------------------------------------------------------
struct TailParams {
    long rows;
    int n;
    const bf16x8 *point_feat;
    const float *gshift;                 // (rows/n, 512)
    const bf16x8 *ws;
    const float *sh7, *sh8, *sh9;
    const float *w10, *b10;              // (128,2), (2)
    float *logits;                       // (rows,2)
    long copy_vecs;                      // > 0: one stream (and one w10) per frustum
    const unsigned *drop;                // (rows,4) keep bits of conv9's outputs, or NULL
};

template <class Feed>
__global__ __launch_bounds__(64 * kWaves) void seg_tail_bf16_kernel(const TailParams pin)
{
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const long row0 = ((long)logical_block(pin.copy_vecs) * kWaves + (threadIdx.x >> 6)) * 32;
    if (row0 >= pin.rows) return;       // (never with a cooperative feed: the host launches it on full workgroups only)
    TailParams p = pin;
    if (pin.copy_vecs) {
        const size_t f = scalar_index(row0 / pin.n);
        p.ws += f * pin.copy_vecs;
        p.w10 += f * (128 * 2);
    }

    // Measured with the stamps below (LDS feed, cycles of one wave, MFMA time in brackets): inputs and first
    // units 1.1 k, conv6 8.3 k [2.0 k], conv7 14.1 k [8.2 k], conv8 4.2 k [2.0 k], conv9 2.5 k [1.0 k], conv10
    // and store 2.6 k.  conv6 (two units per pair) waits for each pair's shifts from memory; staging all shift
    // vectors in LDS first took 2 k off conv6 and put 3.7 k onto the start (the wave's first wait for memory
    // then sits before any MFMA): no gain, not kept.
#ifdef FPN_TAIL_STAMPS      // timing-only build: cycle stamps of one wave into the first logits (wrong results)
    long long stamp[7];
    stamp[0] = clock64();
#define FPN_TSTAMP(i) stamp[i] = clock64()
#else
#define FPN_TSTAMP(i)
#endif
    Feed feed;
    feed.start(p.ws, lane, feed_ring<Feed>());
    Act pf[1][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) pf[0][t].m[s] = p.point_feat[(row0 + pt) * 8 + (2 * t + s) * 2 + h];

    constexpr int U6 = units_of(64, 512), U7 = units_of(512, 256), U8 = units_of(256, 128);
    const bf16x8 *ws = p.ws;
    Act a6[1][16], a7[1][8], a8[1][4];
    floatx16 a9[1][4];
    FPN_TSTAMP(1);
    layer_t_acc<1, 2, 16, 0>(a6, pf, ws, p.gshift + (row0 / p.n) * 512, feed, h);
    FPN_TSTAMP(2);
    ws += U6 * kUnitVecs;
    layer_t_acc<1, 16, 8, U6 % kRing>(a7, a6, ws, p.sh7, feed, h);
    FPN_TSTAMP(3);
    ws += U7 * kUnitVecs;
    layer_t_acc<1, 8, 4, (U6 + U7) % kRing>(a8, a7, ws, p.sh8, feed, h);
    FPN_TSTAMP(4);
    ws += U8 * kUnitVecs;
    layer_t_acc<1, 4, 4, (U6 + U7 + U8) % kRing>(a9, a8, ws, p.sh9, feed, h);
    FPN_TSTAMP(5);

    // Monte-Carlo dropout before the head (models/FPointNet_Bayes.py:92): bit c of word j
    // of the point's mask keeps channel 32 j + c, kept activations count twice
    if (p.drop) {
        const uint4 keep = *reinterpret_cast<const uint4 *>(p.drop + (row0 + pt) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned word = (j == 0 ? keep.x : j == 1 ? keep.y : j == 2 ? keep.z : keep.w) >> (4 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) a9[0][j][r] *= (word >> (8 * (r >> 2) + (r & 3))) & 1u ? 2.f : 0.f;
        }
    }
    // conv10 (128 -> 2, no batch norm, no ReLU) on the vector ALU in fp32: each lane half
    // sums its 64 channels, the halves meet through one cross-lane exchange
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 *w = reinterpret_cast<const float4 *>(p.w10 + (32 * j + 8 * q + 4 * h) * 2);
            const float4 wa = w[0], wb = w[1];
            s0 = fmaf(a9[0][j][4 * q + 0], wa.x, s0); s1 = fmaf(a9[0][j][4 * q + 0], wa.y, s1);
            s0 = fmaf(a9[0][j][4 * q + 1], wa.z, s0); s1 = fmaf(a9[0][j][4 * q + 1], wa.w, s1);
            s0 = fmaf(a9[0][j][4 * q + 2], wb.x, s0); s1 = fmaf(a9[0][j][4 * q + 2], wb.y, s1);
            s0 = fmaf(a9[0][j][4 * q + 3], wb.z, s0); s1 = fmaf(a9[0][j][4 * q + 3], wb.w, s1);
        }
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    if (h == 0) *reinterpret_cast<float2 *>(p.logits + (row0 + pt) * 2) = make_float2(s0 + p.b10[0], s1 + p.b10[1]);
#ifdef FPN_TAIL_STAMPS
    stamp[6] = clock64();
    if (blockIdx.x == gridDim.x / 2 + 8 && threadIdx.x == 64)
        for (int i = 0; i < 7; ++i) reinterpret_cast<long long *>(p.logits)[32 + i] = stamp[i];
#endif
}

// ---------------------------------------------------------------------------
// seg tail, conv6 and conv7 INTERLEAVED.  conv6 is a short layer (two units per pair of
// output tiles): run on its own it waits for every pair's shifts and pays an epilogue per
// eight MFMAs - 8.3 k cycles of a wave for 2.0 k of matrix time - and its 512-wide result
// holds 128 registers per point group until conv7 is done.  Here conv6 is produced in
// chunks of 64 channels (one pair) that conv7's accumulators - all eight output tiles live
// at once - consume at once, and a chunk's units and epilogues ride between conv7's:
//
//     A A | B0 B0 B0 B0 [A A] B1 B1 B1 B1 | B0 B0 B0 B0 [A A] B1 B1 B1 B1 | ...        (8 times)
//
//   A  = a unit of conv6, chunk c+1 (input tile 0 / 1 of point_feat -> acc6),
//   Bj = a unit of conv7 that multiplies tile j of chunk c (one unit per pair of conv7's
//        output tiles); while the B1 units issue, tile 0 of chunk c+1 is finished into the
//        register tile the B0 units have just released, and tile 1 during the next B0 units.
// No epilogue of conv6 is exposed, and the registers it frees make room for NP = 2 point
// groups per wave: every weight fragment is used twice, which halves what the feed has to
// deliver per MFMA.  The shifts of conv6 and conv7 are the accumulators' starting values
// (read from the wave's LDS one unit before the tile's first MFMA), so an epilogue is
// conversion and clamp alone: a third fewer vector instructions next to the MFMAs, 4.5 % of
// the kernel's time.  Every accumulator starts from the same value and sees the same MFMAs
// in the same order as in seg_tail_bf16_kernel (conv7 still adds its input tiles in ascending
// order): the logits are the same bits.  The stream is the same too; the kernel reads its
// units in the order above (TailOrder::pos).
// ---------------------------------------------------------------------------
// P: pairs of conv7's output tiles per pass (4: all eight tiles at once; 2: two passes over conv6 - its 64
// MFMAs per point group are then done twice, 15 % more matrix work, and conv7's live accumulators halve)
template <int P>
struct TailOrder {
    static constexpr int kU6 = 16, kStream = 80;      // conv6: 16 units, conv6 + conv7: 80 units of the stream
    static constexpr int kStep = 2 * P + 2;           // units of a step: P x B0, A A, P x B1 (the last step has no A)
    static constexpr int kPass = 2 + 7 * kStep + 2 * P, kPasses = 4 / P, kFused = kPasses * kPass;
    // what the unit consumed s-th is: kind 0 = A (conv6), 1 = B0, 2 = B1; idx = input tile of A / pair of B inside
    // the pass; c = step (-1: the prologue's A A)
    struct What { int pass, c, kind, idx; };
    static constexpr What what(int S)
    {
        const int pass = S / kPass, s = S % kPass;
        if (s < 2) return {pass, -1, 0, s};
        const int t = s - 2;
        if (t < 7 * kStep) {
            const int c = t / kStep, r = t % kStep;
            if (r < P) return {pass, c, 1, r};
            if (r < P + 2) return {pass, c, 0, r - P};
            return {pass, c, 2, r - P - 2};
        }
        const int q = t - 7 * kStep;
        return q < P ? What{pass, 7, 1, q} : What{pass, 7, 2, q - P};
    }
    // stream position of the unit consumed s-th
    static constexpr int pos(int S)
    {
        if (S >= kFused) return kStream + (S - kFused);
        const What w = what(S);
        if (w.kind == 0) return 2 * (w.c + 1) + w.idx;                   // conv6, chunk c + 1
        return kU6 + (w.pass * P + w.idx) * 16 + 2 * w.c + (w.kind - 1);  // conv7, pair, input tile 2 c (+ 1)
    }
};

// One round of the fused tail: the wave's 32 NP points (conv6's operand fragments in pf, rows row0 ..) through
// conv6 .. conv10; the constants in the wave's LDS block sh6 (offsets TailLds), the dropout bits in keep_w.
// On entry the ring holds the units consumed first, on exit the read-ahead of conv9's last units (the padding
// behind the stream, or with a wrapping feed the next round's first units).  between(): called before conv9.
struct TailLds {
    static constexpr int kSh7 = 512, kSh8 = kSh7 + 256, kSh9 = kSh8 + 128, kW10 = kSh9 + 128, kB10 = kW10 + 256, kFloats = kB10 + 4;
};

template <int NP, int P, class Feed, class Between>
__device__ __forceinline__ void tail_round(const Act (&pf)[NP][2], const float *sh6, const uint4 *keep_w, bool drop, bf16x8 *park,
                                           const bf16x8 *ws, Feed &feed, float *logits, long row0, unsigned lane,
                                           long long *stamp, Between &&between)
{
    using Order = TailOrder<P>;
    constexpr bool kPark = Order::kPasses == 2;
    const int pt = lane & 31, h = lane >> 5;
    const float *sh7 = sh6 + TailLds::kSh7, *sh8 = sh6 + TailLds::kSh8, *sh9 = sh6 + TailLds::kSh9, *w10 = sh6 + TailLds::kW10;
    constexpr int kB10 = TailLds::kB10;
    (void)stamp;
    Act a7[NP][8];
    {
        floatx16 acc7[P][2][NP];
        floatx16 acc6[2][NP];
        Act a6c[NP][2];
        // the shifts enter as the accumulators' starting values (unit_mfma_from): both tiles' shifts of conv6's
        // chunk, or of a conv7 pair, are read from LDS during the unit before the one that starts from them
        TileConst cin[2];
        auto load_pair = [&](const float *sh, int tile0) {
            load_const(cin[0], sh, 32 * tile0 + 4 * h);
            load_const(cin[1], sh, 32 * (tile0 + 1) + 4 * h);
        };
        static_for<0, Order::kFused>([&](auto I) {
            constexpr int S = decltype(I)::value, slot = S % Feed::kDepth;
            constexpr typename Order::What w = Order::what(S);
            constexpr bool first7 = w.c == 0 && w.kind == 1;        // the unit a conv7 pair starts with
            if constexpr (w.kind == 0 && w.idx == 0 && w.c < 0) load_pair(sh6, 0);      // (a pass's opening: nothing to hide it behind)
            if constexpr (w.kind == 0 && w.idx == 0) unit_mfma_from<2>(acc6, feed.template cur<slot>(), pf, cin);
            else if constexpr (w.kind == 0) unit_mfma<false, true, w.idx>(acc6, feed.template cur<slot>(), pf);
            else if constexpr (first7) unit_mfma_from<2>(acc7[w.idx], feed.template cur<slot>(), a6c, cin);
            else unit_mfma<false, true, w.kind - 1>(acc7[w.idx], feed.template cur<slot>(), a6c);
            feed.template advance_at<slot>(ws + (size_t)Order::pos(S + Feed::kLook) * kUnitVecs);
            // riders.  During the B0 units of step c tile 1 of chunk c is finished (one point group per unit), during
            // its B1 units tile 0 of chunk c + 1; in a pass's prologue tile 0 of chunk 0 right behind the second A
            // unit - the one epilogue of conv6 that nothing hides.
            constexpr bool ride1 = w.kind == 1 && w.idx < NP;                   // tile 1 of chunk c, group idx
            constexpr bool ride0 = w.kind == 2 && w.idx < NP && w.c < 7;        // tile 0 of chunk c + 1, group idx
            if constexpr (ride1) finish0(a6c[w.idx][1], acc6[1][w.idx]);
            if constexpr (ride0) finish0(a6c[w.idx][0], acc6[0][w.idx]);
            // the starting values of the next unit
            if constexpr (w.kind == 0 && w.idx == 1 && w.c < 0) load_pair(sh7, 2 * (w.pass * P));                         // conv7, the pass's pair 0
            if constexpr (first7 && w.idx < P - 1) load_pair(sh7, 2 * (w.pass * P + w.idx + 1));                          // ... its next pair
            if constexpr (w.kind == 1 && w.idx == P - 1 && w.c < 7) load_pair(sh6, 2 * (w.c + 1));                        // conv6, chunk c + 1
            sched_unit<Feed::kMode, 4 * NP, 4, (ride0 || ride1) ? (20 + 4 * NP - 1) / (4 * NP) : 1>();
            if constexpr (w.kind == 0 && w.idx == 1 && w.c < 0) {
#pragma unroll
                for (int g = 0; g < NP; ++g) finish0(a6c[g][0], acc6[0][g]);
                __builtin_amdgcn_sched_barrier(0);
            }
            feed.template sync<slot>();
            // a pass's end: conv7's epilogue for its tiles
            if constexpr (S % Order::kPass == Order::kPass - 1) {
#pragma unroll
                for (int tp = 0; tp < P; ++tp)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int g = 0; g < NP; ++g) finish0(a7[g][2 * (w.pass * P + tp) + u], acc7[tp][u][g]);
                if constexpr (kPark && w.pass == 0) {
#pragma unroll
                    for (int g = 0; g < NP; ++g)
#pragma unroll
                        for (int t = 0; t < 2 * P; ++t)
#pragma unroll
                            for (int m = 0; m < 2; ++m) park[((g * 2 * P + t) * 2 + m) * 64] = a7[g][t].m[m];
                }
                __builtin_amdgcn_sched_barrier(0);
                FPN_TSTAMP(2 + (w.pass == Order::kPasses - 1 ? 1 : 0));
            }
        });
        if constexpr (kPark) {
#pragma unroll
            for (int g = 0; g < NP; ++g)
#pragma unroll
                for (int t = 0; t < 2 * P; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m) a7[g][t].m[m] = park[((g * 2 * P + t) * 2 + m) * 64];
        }
    }

    constexpr int U8 = units_of(256, 128);
    ws += (size_t)Order::kStream * kUnitVecs;
    Act a8[NP][4];
    floatx16 a9[NP][4];
    FPN_TSTAMP(4);
    layer_t_acc<NP, 8, 4, Order::kFused % Feed::kDepth>(a8, a7, ws, sh8, feed, h);
    FPN_TSTAMP(5);
    ws += U8 * kUnitVecs;
    between();       // (the caller's loads for its next round: conv9 and conv10 are time enough for them to arrive)
    layer_t_acc<NP, 4, 4, (Order::kFused + U8) % Feed::kDepth>(a9, a8, ws, sh9, feed, h);
    FPN_TSTAMP(6);

#pragma unroll
    for (int g = 0; g < NP; ++g) {
        // Monte-Carlo dropout before the head (models/FPointNet_Bayes.py:92): bit c of word j
        // of the point's mask keeps channel 32 j + c, kept activations count twice
        if (drop) {
            const uint4 keep = keep_w[g * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned word = (j == 0 ? keep.x : j == 1 ? keep.y : j == 2 ? keep.z : keep.w) >> (4 * h);
#pragma unroll
                for (int r = 0; r < 16; ++r) a9[g][j][r] *= (word >> (8 * (r >> 2) + (r & 3))) & 1u ? 2.f : 0.f;
            }
        }
    }
    // conv10 (128 -> 2) in fp32 on the vector ALU, as in seg_tail_bf16_kernel: per point group the same fused
    // multiply-adds in the same order.  Both logits of a channel in one packed instruction (no MFMA runs beside
    // these, where a packed instruction would cost more than its two halves), the point groups side by side - one
    // LDS read of a weight pair serves both, and their sums are independent chains - and the bias from LDS: read
    // from memory here, it was a whole trip to memory with nothing else to do.
    floatx2 s01[NP];
#pragma unroll
    for (int g = 0; g < NP; ++g) s01[g] = floatx2{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const floatx2 *w = reinterpret_cast<const floatx2 *>(w10 + (32 * j + 8 * q + 4 * h) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < NP; ++g) {
                    const float a = a9[g][j][4 * q + e];
                    s01[g] = __builtin_elementwise_fma(floatx2{a, a}, w[e], s01[g]);
                }
        }
    const floatx2 b10 = *reinterpret_cast<const floatx2 *>(sh6 + kB10);
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        float s0 = s01[g][0], s1 = s01[g][1];
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (h == 0) *reinterpret_cast<float2 *>(logits + (row0 + 32 * g + pt) * 2) = make_float2(s0 + b10[0], s1 + b10[1]);
    }
}

template <int NP, int P, class Feed>
__global__ __launch_bounds__(64 * kWaves) void seg_tail_fused_bf16_kernel(const TailParams pin)
{
    using Order = TailOrder<P>;
    static_assert(NP <= P, "a tile's riders need one unit per point group");
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const long row0 = ((long)logical_block(pin.copy_vecs) * kWaves + (threadIdx.x >> 6)) * (32 * NP);
    // conv6's shifts - the frustum's row of 512 floats - wait in the wave's OWN piece of LDS (each wave reads only
    // what it wrote: no barrier) and come back a chunk's two tiles at a time, one unit before the MFMAs that
    // start from them
    // The other constants of the wave's layers wait there too (conv7's, conv8's and conv9's shifts, conv10's
    // matrix, behind the 512 floats of conv6: see StagedConsts; read from memory they cost 6 % of the kernel).
    constexpr int kSh7 = TailLds::kSh7, kSh8 = TailLds::kSh8, kW10 = TailLds::kW10, kB10 = TailLds::kB10;
    __shared__ float shift6[kWaves][TailLds::kFloats];
    // two passes: the first pass's half of conv7's output (64 registers of operand fragments for two point
    // groups) has nothing to do during the second pass and waits in the wave's own piece of LDS as well
    // (vector v of a lane at [v][lane]: consecutive 16-byte slots, no bank conflicts) - held in registers it
    // pushed the compiler into parking values in accumulator registers and copying whole accumulator tiles around
    constexpr bool kPark = Order::kPasses == 2;
    constexpr int kParkVecs = NP * P * 2 * 2;        // point groups x tiles of a pass x fragments
    __shared__ bf16x8 park7[kPark ? kWaves * kParkVecs * 64 : 1];
    if (row0 >= pin.rows) return;       // (never with a cooperative feed: the host launches it on full workgroups only)
    TailParams p = pin;
    if (pin.copy_vecs) {
        const size_t f = scalar_index(row0 / pin.n);
        p.ws += f * pin.copy_vecs;
        p.w10 += f * (128 * 2);
    }
    const bf16x8 *ws = p.ws;
#ifdef FPN_TAIL_STAMPS      // timing-only build: cycle stamps of one wave into the first logits (wrong results)
    long long stamp[8];
    stamp[0] = clock64();
#endif
    Feed feed;
    feed.start_at([&](auto I) { return ws + (size_t)Order::pos(decltype(I)::value) * kUnitVecs; }, lane, feed_ring<Feed>(), ws);
    bf16x8 *park = park7 + (kPark ? (threadIdx.x >> 6) * (kParkVecs * 64) + lane : 0);

    // the opening's loads, all requested before the first LDS write (a write right behind its load would make
    // the wave wait there for everything asked so far, and ask for the rest only then): the conv6 shifts of
    // the frustum, the later layers' constants, the points' dropout bits - wanted at the very end: requested
    // there, they would be waited for with nothing left to compute - and conv6's operands
    float *sh6 = shift6[threadIdx.x >> 6];
    __shared__ uint4 keep_bits[kWaves][NP][64];
    uint4 *keep_w = keep_bits[threadIdx.x >> 6][0];
    Act pf[NP][2];
    {
        typedef float vec4 __attribute__((ext_vector_type(4)));
        typedef unsigned uvec4 __attribute__((ext_vector_type(4)));
        const vec4 *src = reinterpret_cast<const vec4 *>(p.gshift + (row0 / p.n) * 512);   // (a wave's points lie in one frustum)
        const vec4 lo = src[lane], hi = src[64 + lane];
        const vec4 c7 = reinterpret_cast<const vec4 *>(p.sh7)[lane];
        const vec4 c89 = reinterpret_cast<const vec4 *>(lane < 32 ? p.sh8 : p.sh9)[lane & 31];
        const vec4 c10 = reinterpret_cast<const vec4 *>(p.w10)[lane];
        const float cb10 = p.b10[lane & 1];
        // (without dropout the same loads read the start of conv7's shifts instead, and nothing reads what they
        // put in LDS: a branch round them made the compiler wait for every load before it on the other path)
        uvec4 keep[NP];
#pragma unroll
        for (int g = 0; g < NP; ++g)
            keep[g] = *reinterpret_cast<const uvec4 *>(p.drop ? p.drop + (row0 + 32 * g + pt) * 4 : reinterpret_cast<const unsigned *>(p.sh7));
#pragma unroll
        for (int g = 0; g < NP; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) pf[g][t].m[s] = p.point_feat[(row0 + 32 * g + pt) * 8 + (2 * t + s) * 2 + h];
        __builtin_amdgcn_sched_barrier(0);
        reinterpret_cast<vec4 *>(sh6)[lane] = lo;
        reinterpret_cast<vec4 *>(sh6)[64 + lane] = hi;
        reinterpret_cast<vec4 *>(sh6 + kSh7)[lane] = c7;
        reinterpret_cast<vec4 *>(sh6 + kSh8)[lane] = c89;           // (conv8's 128 shifts, then conv9's)
        reinterpret_cast<vec4 *>(sh6 + kW10)[lane] = c10;
        if (lane < 2) sh6[kB10 + lane] = cb10;
#pragma unroll
        for (int g = 0; g < NP; ++g) reinterpret_cast<uvec4 *>(keep_w)[g * 64 + lane] = keep[g];
    }
    FPN_TSTAMP(1);
#ifdef FPN_TAIL_STAMPS
    long long *stamps = stamp;
#else
    long long *stamps = nullptr;
#endif
    tail_round<NP, P>(pf, sh6, keep_w, p.drop != nullptr, park, ws, feed, p.logits, row0, lane, stamps, [] {});
#ifdef FPN_TAIL_STAMPS
    stamp[7] = clock64();
    if (blockIdx.x == gridDim.x / 2 + 8 && threadIdx.x == 64)
        for (int i = 0; i < 8; ++i) reinterpret_cast<long long *>(p.logits)[32 + i] = stamp[i];
#endif
}

// ---------------------------------------------------------------------------
// The fused tail with a WORKGROUP PER FRUSTUM (or fewer workgroups that walk the frustums), as the head's
// seg_head_frustum_bf16_kernel: a wave's opening - constants into LDS, the ring's first units, conv6's operands,
// all from memory with nothing else on the SIMD - is 2.3 k of the 45 k cycles of a round.  Here a wave takes
// every kWaves-th round of 32 NP points of one frustum: the constants are staged once per frustum, the ring wraps
// to the units the next round consumes first (TailWrapFeedT), and the next round's operands and dropout bits are
// requested before conv9 and have arrived when conv10 is done.  Same arithmetic per point as
// seg_tail_fused_bf16_kernel: same bits.  Needs n % (kWaves * 32 NP) == 0.
// ---------------------------------------------------------------------------
template <int NP, int P>
__global__ __launch_bounds__(64 * kWaves) void seg_tail_walk_bf16_kernel(const TailParams pin)
{
    using Order = TailOrder<P>;
    using Feed = TailWrapFeedT<kRing>;
    static_assert(NP <= P && (Order::kFused + units_of(256, 128) + units_of(128, 128)) % kRing == 0,
                  "the ring's slots must come round with the stream");
    constexpr int kRound = kWaves * 32 * NP;
    constexpr int kUnits = Order::kStream + units_of(256, 128) + units_of(128, 128);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    __shared__ float shift6[kWaves][TailLds::kFloats];
    constexpr bool kPark = Order::kPasses == 2;
    constexpr int kParkVecs = NP * P * 2 * 2;
    __shared__ bf16x8 park7[kPark ? kWaves * kParkVecs * 64 : 1];
    __shared__ uint4 keep_bits[kWaves][NP][64];
    float *sh6 = shift6[wave];
    uint4 *keep_w = keep_bits[wave][0];
    bf16x8 *park = park7 + (kPark ? wave * (kParkVecs * 64) + lane : 0);
    typedef float vec4 __attribute__((ext_vector_type(4)));
    typedef unsigned uvec4 __attribute__((ext_vector_type(4)));
    const long frustums = pin.rows / pin.n;
    Feed feed;
#pragma unroll
    for (int i = 0; i < kRing; ++i) feed.first4[i] = Order::pos(i);

    auto request = [&](Act (&pf)[NP][2], uvec4 (&keep)[NP], long row0) {     // a round's inputs (unconditional loads)
#pragma unroll
        for (int g = 0; g < NP; ++g)
            keep[g] = *reinterpret_cast<const uvec4 *>(pin.drop ? pin.drop + (row0 + 32 * g + pt) * 4 : reinterpret_cast<const unsigned *>(pin.sh7));
#pragma unroll
        for (int g = 0; g < NP; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) pf[g][t].m[s] = pin.point_feat[(row0 + 32 * g + pt) * 8 + (2 * t + s) * 2 + h];
    };

#pragma unroll 1
    for (long f = gridDim.x == frustums ? (long)logical_block(pin.copy_vecs) : (long)blockIdx.x; f < frustums; f += gridDim.x) {
        const bf16x8 *ws = pin.ws + (pin.copy_vecs ? (size_t)f * pin.copy_vecs : 0);
        const float *w10g = pin.w10 + (pin.copy_vecs ? (size_t)f * (128 * 2) : 0);
        feed.base = ws;
        feed.end = ws + (size_t)kUnits * kUnitVecs;
        feed.start_at([&](auto I) { return ws + (size_t)Order::pos(decltype(I)::value) * kUnitVecs; }, lane, nullptr, ws);
        long row0 = f * pin.n + (long)wave * (32 * NP);
        const long row_end = (f + 1) * pin.n;
        Act pf[NP][2];
        {
            const vec4 *src = reinterpret_cast<const vec4 *>(pin.gshift + f * 512);
            const vec4 lo = src[lane], hi = src[64 + lane];
            const vec4 c7 = reinterpret_cast<const vec4 *>(pin.sh7)[lane];
            const vec4 c89 = reinterpret_cast<const vec4 *>(lane < 32 ? pin.sh8 : pin.sh9)[lane & 31];
            const vec4 c10 = reinterpret_cast<const vec4 *>(w10g)[lane];
            const float cb10 = pin.b10[lane & 1];
            uvec4 keep[NP];
            request(pf, keep, row0);
            __builtin_amdgcn_sched_barrier(0);
            reinterpret_cast<vec4 *>(sh6)[lane] = lo;
            reinterpret_cast<vec4 *>(sh6)[64 + lane] = hi;
            reinterpret_cast<vec4 *>(sh6 + TailLds::kSh7)[lane] = c7;
            reinterpret_cast<vec4 *>(sh6 + TailLds::kSh8)[lane] = c89;
            reinterpret_cast<vec4 *>(sh6 + TailLds::kW10)[lane] = c10;
            if (lane < 2) sh6[TailLds::kB10 + lane] = cb10;
#pragma unroll
            for (int g = 0; g < NP; ++g) reinterpret_cast<uvec4 *>(keep_w)[g * 64 + lane] = keep[g];
        }
#pragma unroll 1
        for (;;) {
            const long next = row0 + kRound;
            const bool more = next < row_end;
            Act pfn[NP][2];
            uvec4 keepn[NP];
            tail_round<NP, P>(pf, sh6, keep_w, pin.drop != nullptr, park, ws, feed, pin.logits, row0, lane, nullptr,
                              [&] { request(pfn, keepn, more ? next : row0); });
            if (!more) break;
#pragma unroll
            for (int g = 0; g < NP; ++g) {
                reinterpret_cast<uvec4 *>(keep_w)[g * 64 + lane] = keepn[g];
#pragma unroll
                for (int t = 0; t < 2; ++t) pf[g][t] = pfn[g][t];
            }
            row0 = next;
        }
    }
}

// ---------------------------------------------------------------------------
// object-point stacks.  Stream: layer 2, layer 3 (, layer 4).
// ---------------------------------------------------------------------------
struct ObjParams {
    long rows;
    int m;
    const float *obj, *delta;
    const float *w1, *sc1, *sh1;       // (3,128) fp32
    const bf16x8 *ws;
    const float *sh2, *sh3, *sh4;
    float *gmax;
    int ldg;
    long copy_vecs;                    // > 0: one stream (and one w1) per frustum
};

template <bool BOX, int NP, class Feed = RingFeed>
__global__ __launch_bounds__(64 * kWaves) void obj_chain_bf16_kernel(const ObjParams pin)
{
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const unsigned block = logical_block(pin.copy_vecs);
    const long row0 = ((long)block * kWaves + (threadIdx.x >> 6)) * (32 * NP);
    constexpr int C = BOX ? 512 : 256;
    __shared__ int lmax[4 * C];
    const WgPool pool = wg_pool<NP>(pin.rows, pin.m, block);
    if (!pool.on && row0 >= pin.rows) return;
    ObjParams p = pin;
    if (pin.copy_vecs) {
        const size_t f = scalar_index(row0 / pin.m);
        p.ws += f * pin.copy_vecs;
        p.w1 += f * (3 * 128);
    }
    if (pool.on) wg_pool_clear<C>(lmax, pool);
    const long frustum = row0 / p.m;
    int *gmax = reinterpret_cast<int *>(p.gmax + frustum * p.ldg);

    Feed feed;
    feed.start(p.ws, lane, feed_ring<Feed>());
    __shared__ float oconst[kWaves][128 + 256 + (BOX ? 512 : 0)];
    float *lc = oconst[threadIdx.x >> 6];
    StagedConsts<128> c2;
    StagedConsts<256> c3;
    StagedConsts<BOX ? 512 : 4> c4;
    c2.request(p.sh2, lane);
    c3.request(p.sh3, lane);
    if constexpr (BOX) c4.request(p.sh4, lane);
    const float *sh2 = lc, *sh3 = lc + 128, *sh4 = lc + 384;
    Act a1[NP][4], a2[NP][4];
    {
        FirstLayer<4, 4> first;
        first.load(p.w1, 3, p.sc1, p.sh1, lane, h);
        float x[NP][4];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const float *src = p.obj + (row0 + 32 * g + pt) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) x[g][k] = src[k] - (p.delta ? p.delta[frustum * 3 + k] : 0.f);
            x[g][3] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);      // (the constants' writes stay behind the requests above)
        c2.put(lc, lane);
        c3.put(lc + 128, lane);
        if constexpr (BOX) c4.put(lc + 384, lane);
        first.apply(a1, x, h);
    }
    constexpr int U2 = units_of(128, 128), U3 = units_of(128, 256);
    const bf16x8 *ws = p.ws;
    layer_t_acc<NP, 4, 4, 0>(a2, a1, ws, sh2, feed, h);
    ws += U2 * kUnitVecs;
    if constexpr (!BOX) {
        if (pool.on) layer_max<NP, 4, 8, U2 % kRing>(a2, ws, sh3, lmax + pool.local * C, feed, lane);
        else layer_max<NP, 4, 8, U2 % kRing>(a2, ws, sh3, gmax, feed, lane);
    } else {
        Act a3[NP][8];
        layer_t_acc<NP, 4, 8, U2 % kRing>(a3, a2, ws, sh3, feed, h);
        ws += U3 * kUnitVecs;
        if (pool.on) layer_max<NP, 8, 16, (U2 + U3) % kRing>(a3, ws, sh4, lmax + pool.local * C, feed, lane);
        else layer_max<NP, 8, 16, (U2 + U3) % kRing>(a3, ws, sh4, gmax, feed, lane);
    }
    if (pool.on) wg_pool_flush<C>(lmax, pool, p.gmax, p.ldg, C);
}

// ---------------------------------------------------------------------------
// The object-point stacks with workgroups that WALK the frustums (one stream for all frustums), as the seg
// head's and tail's: obj_chain_bf16_kernel's waves live for one round of 32 NP points and open every round with
// a trip to memory - the first weight units, the first layer's matrix and constants, the layers' shifts, the
// points - which is a fifth of a T-Net round.  Here the first layer's operands, the shifts (one LDS copy for
// the workgroup) and the ring are set up once per workgroup; the waves take every kWaves-th round of a frustum
// and then the next frustum, over a wrapping ring; the next round's points are requested as soon as this
// round's have been multiplied; a frustum's maxima meet in LDS and leave with one atomic per channel.  Same
// arithmetic per point as obj_chain_bf16_kernel: same bits.  Needs m % (kWaves * 32 NP) == 0.
// ---------------------------------------------------------------------------
template <bool BOX, int NP>
__global__ __launch_bounds__(64 * kWaves) void obj_chain_walk_bf16_kernel(const ObjParams p)
{
    using Feed = WrapRingFeedT<kRing>;
    constexpr int kRound = kWaves * 32 * NP, C = BOX ? 512 : 256;
    constexpr int U2 = units_of(128, 128), U3 = units_of(128, 256), U4 = BOX ? units_of(256, 512) : 0, U = U2 + U3 + U4;
    static_assert(U % kRing == 0, "the ring's slots must come round with the stream");
    constexpr int kConsts = 128 + 256 + (BOX ? 512 : 0);
    static_assert(kConsts / 4 <= 64 * kWaves, "one vector of constants a thread");
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    __shared__ int lmax[C];
    __shared__ float oconst[kConsts];
    const long frustums = p.rows / p.m;

    Feed feed;
    feed.start(p.ws, lane, nullptr);
    feed.end = p.ws + (size_t)U * kUnitVecs;
    feed.len = (long)U * kUnitVecs;
    FirstLayer<4, 4> first;
    first.load(p.w1, 3, p.sc1, p.sh1, lane, h);
    // a round's points minus the frustum's delta (unconditional loads: without a delta the points stand in)
    auto request = [&](float (&x)[NP][4], long row0, long frustum) {
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const float *src = p.obj + (row0 + 32 * g + pt) * 3;
            const float *dl = p.delta ? p.delta + frustum * 3 : src;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float v = src[k], d = dl[k];
                x[g][k] = p.delta ? v - d : v;
            }
            x[g][3] = 0.f;
        }
    };
    long f = blockIdx.x;
    float x[NP][4];
    request(x, (f < frustums ? f : 0) * p.m + (long)wave * (32 * NP), f < frustums ? f : 0);
    {
        typedef float vec4 __attribute__((ext_vector_type(4)));
        const int k = (int)threadIdx.x < kConsts / 4 ? (int)threadIdx.x : 0;      // sh2's 32 vectors, sh3's 64, sh4's 128
        const float *base = k < 32 ? p.sh2 : k < 96 || !BOX ? p.sh3 - 128 : p.sh4 - 384;
        const vec4 v = reinterpret_cast<const vec4 *>(base)[k];
        __builtin_amdgcn_sched_barrier(0);
        if ((int)threadIdx.x < kConsts / 4) reinterpret_cast<vec4 *>(oconst)[threadIdx.x] = v;
    }
    const float *sh2 = oconst, *sh3 = oconst + 128, *sh4 = oconst + 384;
    const bf16x8 *ws = p.ws;

#pragma unroll 1
    for (; f < frustums; f += gridDim.x) {
        for (int i = threadIdx.x; i < C; i += 64 * kWaves) lmax[i] = 0;
        __syncthreads();        // (also: the constants are there; the flush before read only what this thread clears)
#pragma unroll 1
        for (int r0 = (int)wave * (32 * NP); r0 < p.m; r0 += kRound) {
            Act a1[NP][4], a2[NP][4];
            first.apply(a1, x, h);
            {       // the next round's points: of this frustum, or the first of the workgroup's next one
                const bool here = r0 + kRound < p.m;
                const long fn = here ? f : (f + gridDim.x < frustums ? f + gridDim.x : f);
                request(x, fn * p.m + (here ? r0 + kRound : (int)wave * (32 * NP)), fn);
            }
            layer_t_acc<NP, 4, 4, 0>(a2, a1, ws, sh2, feed, h);
            if constexpr (!BOX) {
                layer_max<NP, 4, 8, U2 % kRing>(a2, ws + U2 * kUnitVecs, sh3, lmax, feed, lane);
            } else {
                Act a3[NP][8];
                layer_t_acc<NP, 4, 8, U2 % kRing>(a3, a2, ws + U2 * kUnitVecs, sh3, feed, h);
                layer_max<NP, 8, 16, (U2 + U3) % kRing>(a3, ws + (U2 + U3) * kUnitVecs, sh4, lmax, feed, lane);
            }
        }
        __syncthreads();
        for (int ch = threadIdx.x; ch < C; ch += 64 * kWaves)
            atomicMax(reinterpret_cast<int *>(p.gmax + f * p.ldg + ch), lmax[ch]);
    }
}

// ---------------------------------------------------------------------------
// Set-abstraction module (models/pointnet_util.py:86-195) as one kernel, as
// sa_chain_kernel of chain_reg.hip; a wave takes 32 NP consecutive samples of one ball.
// NJF > 0: the gathered point features are the first layer's operand tiles (converted to
// bf16 on the way in) and the relative coordinates ride in one more tile; stream: layer 1
// ((NJF+1)*32 rows), layer 2, layer 3.  NJF = 0: first layer in fp32 from memory; stream:
// layer 2, layer 3.
// ---------------------------------------------------------------------------
#ifndef FPN_SA_OCC
#define FPN_SA_OCC(njf) 1
#endif
struct SaParams {
    long rows;
    int n, m, ns, cf;
    const float *xyz, *new_xyz, *feat;
    const bf16x8 *featp;                // NJF > 0: the features packed by fpn_sa_pack_features_bf16, or NULL (then feat)
    const float *pre;                   // NJF = 0: (b*n, ldp >= 32 NT1) what wide features contribute to the first layer, or NULL
    int ldp;
    const int *idx;
    const float *w1, *sc1, *sh1;        // NJF = 0 only: (3+cf, 32 NT1) fp32, scale, shift
    const bf16x8 *ws;
    const float *shs1, *sh2, *sh3;      // shifts of the streamed layers (shs1: NJF > 0 only)
    float *out;
    int ldo, c3;
};

template <int NJF, int NT1, int NT2, int NT3, int NP, int KP = 8, class Feed = RingFeed>
__global__ __launch_bounds__(64 * kWaves, FPN_SA_OCC(NJF)) void sa_chain_bf16_kernel(const SaParams p)
{
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const unsigned block = logical_block(1);      // a frustum's balls gather from the same arrays: one XCD
    const long row0 = ((long)block * kWaves + (threadIdx.x >> 6)) * (32 * NP);
    constexpr int C = 32 * NT3;
    __shared__ int lmax[4 * C];
    const WgPool pool = wg_pool<NP>(p.rows, p.ns, block);
    if (!pool.on && row0 >= p.rows) return;
    // the ball's point indices first: the gathers below wait for them, and loads return in
    // issue order - behind the weight ring's first units they would arrive a ring later
    int pid[NP];
#pragma unroll
    for (int g = 0; g < NP; ++g) pid[g] = p.idx[row0 + 32 * g + pt];
    __builtin_amdgcn_sched_barrier(0);
    if (pool.on) wg_pool_clear<C>(lmax, pool);
    const long ball = row0 / p.ns;
    const long b = ball / p.m;
    const float *ctr = p.new_xyz + ball * 3;
    int *gmax = reinterpret_cast<int *>(p.out + ball * p.ldo);
#ifdef FPN_SA_STAMPS      // timing-only build: cycle stamps of one wave into the first output row (wrong results)
    long long stamp[6];
    stamp[0] = clock64();
#define FPN_STAMP(i) stamp[i] = clock64()
#else
#define FPN_STAMP(i)
#endif

    Feed feed;
    feed.start(p.ws, lane, feed_ring<Feed>());
    const bf16x8 *ws = p.ws;
    Act a1[NP][NT1], a2[NP][NT2];
    if constexpr (NJF > 0) {
        constexpr int NJ1 = NJF + 1;
        Act in[NP][NJ1];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const float *src = p.xyz + (b * p.n + pid[g]) * 3;
            if (p.featp) {      // (kernel-uniform) operand fragments as they lie in memory: half the gathered bytes, no conversion
                const bf16x8 *f = p.featp + (b * p.n + pid[g]) * (long)(4 * NJF);
#pragma unroll
                for (int t = 0; t < NJF; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) in[g][t].m[s] = f[(2 * t + s) * 2 + h];
            } else {
                const float *f = p.feat + (b * p.n + pid[g]) * (long)p.cf;
#pragma unroll
                for (int t = 0; t < NJF; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4 *>(f + 32 * t + 8 * q + 4 * h);
                        in[g][t].m[q >> 1][4 * (q & 1) + 0] = (__bf16)v.x;
                        in[g][t].m[q >> 1][4 * (q & 1) + 1] = (__bf16)v.y;
                        in[g][t].m[q >> 1][4 * (q & 1) + 2] = (__bf16)v.z;
                        in[g][t].m[q >> 1][4 * (q & 1) + 3] = (__bf16)v.w;
                    }
            }
            // the coordinates ride in the last tile as two bf16 numbers each (one alone would quantise
            // metres to decimetres): the high parts in channels 0..2 (elements 0..2 of fragment 0 in the
            // lower lane half), what they leave over in channels 4..6 (the same elements in the upper
            // half), against the same three weight rows
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) in[g][NJF].m[s][e] = (__bf16)0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = src[k] - ctr[k];
                const __bf16 hi = (__bf16)d;
                in[g][NJF].m[0][k] = h == 0 ? hi : (__bf16)(d - (float)hi);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        FPN_STAMP(1);
        constexpr int U1 = (NT1 / 2) * NJ1, U2 = (NT2 / 2) * NT1;
        layer_t<NP, NJ1, NT1, 0>(a1, in, ws, p.shs1, feed, h);
        FPN_STAMP(2);
        ws += U1 * kUnitVecs;
        layer_t<NP, NT1, NT2, U1 % kRing>(a2, a1, ws, p.sh2, feed, h);
        FPN_STAMP(3);
        ws += U2 * kUnitVecs;
        if (pool.on) layer_max<NP, NT2, NT3, (U1 + U2) % kRing>(a2, ws, p.sh3, lmax + pool.local * C, feed, lane, p.c3);
        else layer_max<NP, NT2, NT3, (U1 + U2) % kRing>(a2, ws, p.sh3, gmax, feed, lane, p.c3);
        FPN_STAMP(4);
    } else {
        FirstLayer<NT1, KP> first;
        first.load(p.w1, 3 + p.cf, p.sc1, p.sh1, lane, h);
        float x[NP][KP];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const float *src = p.xyz + (b * p.n + pid[g]) * 3;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float v = 0.f;
                if (k < 3) v = src[k] - ctr[k];
                else if (k < 3 + p.cf) v = p.feat[(b * p.n + pid[g]) * (long)p.cf + k - 3];
                x[g][k] = v;
            }
        }
        FPN_STAMP(1);
        if (p.pre) {            // (kernel-uniform)
            floatx16 init[NP][NT1];
#pragma unroll
            for (int g = 0; g < NP; ++g)
#pragma unroll
                for (int t = 0; t < NT1; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4 *>(p.pre + (b * p.n + pid[g]) * (long)p.ldp + 32 * t +
                                                                           8 * q + 4 * h);
                        init[g][t][4 * q + 0] = v.x; init[g][t][4 * q + 1] = v.y;
                        init[g][t][4 * q + 2] = v.z; init[g][t][4 * q + 3] = v.w;
                    }
            first.apply(a1, x, h, init);
        } else {
            first.apply(a1, x, h);
        }
        FPN_STAMP(2);
        constexpr int U2 = ((NT2 + 1) / 2) * NT1;
        TileConst c2[2];        // (the arithmetic of the persistent kernel, bit for bit)
        load_const(c2[0], p.sh2, 4 * h);
        load_const(c2[1], p.sh2, 32 + 4 * h);
        layer_t_from<NP, NT1, NT2, 0>(a2, a1, ws, p.sh2, feed, h, c2);
        FPN_STAMP(3);
        ws += U2 * kUnitVecs;
        if (pool.on) layer_max<NP, NT2, NT3, U2 % kRing>(a2, ws, p.sh3, lmax + pool.local * C, feed, lane, p.c3);
        else layer_max<NP, NT2, NT3, U2 % kRing>(a2, ws, p.sh3, gmax, feed, lane, p.c3);
        FPN_STAMP(4);
    }
    if (pool.on) wg_pool_flush<C>(lmax, pool, p.out, p.ldo, p.c3);
#ifdef FPN_SA_STAMPS
    stamp[5] = clock64();
    if (blockIdx.x == gridDim.x / 2 + 8 && threadIdx.x == 64)
        for (int i = 0; i < 6; ++i) reinterpret_cast<long long *>(p.out)[i] = stamp[i];
#endif
}

// ---------------------------------------------------------------------------
// The same module when the first layer reads coordinates (NJF = 0: the stacks of the first
// multi-scale level, 29 M gathered rows per 1024 frustums through two small layers), as a
// PERSISTENT kernel.  Launched per group of points, such a wave spent a quarter of its time
// in MFMAs: before them stood the chain idx -> coordinates -> first layer and the first
// units of the weight ring, each a trip to memory that a single in-order wave per SIMD
// cannot hide.  Here a workgroup per CU copies the stream (24-48 KiB) and the shifts into
// LDS once, and each wave then walks whole balls, 32 NP samples per round:
//  * weights and shifts come from LDS (LdsFeed), the only memory reads inside a round are
//    the NEXT round's coordinates and the round after's indices, issued before the round's
//    first MFMA and not waited for until the next round begins;
//  * a ball's maxima meet in the wave's own LDS row (no other wave shares a ball), and the
//    wave stores the finished row itself: no memory atomics, no barrier after the set-up.
// KP: input channels of the first layer, padded (4 or 8).
// ---------------------------------------------------------------------------
#ifndef FPN_SAP_OCC
#define FPN_SAP_OCC(np) ((np) == 1 ? 2 : 1)
#endif
// WPB: waves per workgroup (they share the LDS copy of the stream).
template <int NT1, int NT2, int NT3, int NP, int KP, bool PRE = false, int WPB = kWaves>
__global__ __launch_bounds__(64 * WPB, PRE ? 1 : FPN_SAP_OCC(NP)) void sa_first_bf16_kernel(const SaParams p, unsigned balls,
                                                                                      unsigned balls_per_wave)
{
    constexpr int U2 = ((NT2 + 1) / 2) * NT1, U3 = (NT3 / 2) * NT2, U = U2 + U3, C = 32 * NT3;
    static_assert(U % kLdsAhead == 0, "unsupported stack");
    extern __shared__ bf16x8 sa_first_lds[];
    bf16x8 *wl = sa_first_lds;                                          // U + kLdsAhead units
    float *lsh2 = reinterpret_cast<float *>(wl + (U + kLdsAhead) * kUnitVecs);
    float *lsh3 = lsh2 + 32 * NT2;
    int *lmax = reinterpret_cast<int *>(lsh3 + C) + (threadIdx.x >> 6) * C;
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;

    for (int i = threadIdx.x; i < U * kUnitVecs; i += 64 * WPB) wl[i] = p.ws[i];
    for (int i = threadIdx.x; i < kLdsAhead * kUnitVecs; i += 64 * WPB) wl[U * kUnitVecs + i] = p.ws[i];
    for (int i = threadIdx.x; i < 32 * NT2; i += 64 * WPB) lsh2[i] = p.sh2[i];
    for (int i = threadIdx.x; i < C; i += 64 * WPB) lsh3[i] = p.sh3[i];
    for (int i = lane; i < C; i += 64) lmax[i] = 0;
    __syncthreads();

    const unsigned gw = __builtin_amdgcn_readfirstlane(logical_block(1) * WPB + (threadIdx.x >> 6));
    const unsigned ball0 = gw * balls_per_wave, ball1 = min(balls, ball0 + balls_per_wave);
    if (ball0 >= ball1) return;
    const unsigned cpb = p.ns / (32 * NP);          // rounds per ball
    const unsigned c0 = ball0 * cpb, c1 = ball1 * cpb;

    FirstLayer<NT1, KP> first;
    first.load(p.w1, 3 + p.cf, p.sc1, p.sh1, lane, h);

    int pid[NP];
    float xr[NP][KP], cr[3];
    float4 pr[PRE ? NP : 1][NT1][4];        // PRE: the next round's rows of first-layer products, as they lie in memory
    auto load_idx = [&](unsigned c) {
#pragma unroll
        for (int g = 0; g < NP; ++g) pid[g] = p.idx[(size_t)c * (32 * NP) + 32 * g + pt];
    };
    auto load_x = [&](unsigned c) {         // raw values: nothing here waits for a load it has issued itself
        const unsigned ball = c / cpb, b = ball / p.m;
#pragma unroll
        for (int k = 0; k < 3; ++k) cr[k] = p.new_xyz[(size_t)ball * 3 + k];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const size_t q = (size_t)b * p.n + pid[g];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float v = 0.f;
                if (k < 3) v = p.xyz[q * 3 + k];
                else if (k < 3 + p.cf) v = p.feat[q * p.cf + k - 3];
                xr[g][k] = v;
            }
            if constexpr (PRE) {
#pragma unroll
                for (int t = 0; t < NT1; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        pr[g][t][i] = *reinterpret_cast<const float4 *>(p.pre + q * p.ldp + 32 * t + 8 * i + 4 * h);
            }
        }
    };
    load_idx(c0);
    load_x(c0);
    load_idx(min(c0 + 1, c1 - 1));

    LdsFeed feed;
    feed.start(p.ws, lane, wl);
    const bf16x8 *ws = p.ws;
    unsigned ball = ball0, in_ball = 0;
#ifdef FPN_SA_STAMPS      // timing-only build: one wave's cycles per phase, averaged over its rounds, into the first output row
    long long phase[6] = {0, 0, 0, 0, 0, 0}, last = clock64();
#define FPN_PSTAMP(i) { const long long now = clock64(); phase[i] += now - last; last = now; }
#else
#define FPN_PSTAMP(i)
#endif
#pragma unroll 1
    for (unsigned c = c0; c < c1; ++c) {
        float x[NP][KP];
#pragma unroll
        for (int g = 0; g < NP; ++g)
#pragma unroll
            for (int k = 0; k < KP; ++k) x[g][k] = k < 3 ? xr[g][k] - cr[k] : xr[g][k];
        floatx16 init[PRE ? NP : 1][NT1];
        if constexpr (PRE) {
#pragma unroll
            for (int g = 0; g < NP; ++g)
#pragma unroll
                for (int t = 0; t < NT1; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        init[g][t][4 * i + 0] = pr[g][t][i].x; init[g][t][4 * i + 1] = pr[g][t][i].y;
                        init[g][t][4 * i + 2] = pr[g][t][i].z; init[g][t][4 * i + 3] = pr[g][t][i].w;
                    }
        }
        // The next round's inputs (and the round after's indices, last: the addresses above come from the ones
        // held now) are requested between the second layer's MFMAs, a few per unit: issued in one burst at the top
        // of the round, 13 to 45 gathers of 64 different lines each kept the wave from issuing MFMAs for 0.7-3 k
        // of a round's 3-10 k cycles.
        const unsigned cn = min(c + 1, c1 - 1), cnn = min(c + 2, c1 - 1);
        const unsigned balln = cn / cpb, bn = balln / p.m;
        size_t qn[NP];
#pragma unroll
        for (int g = 0; g < NP; ++g) qn[g] = (size_t)bn * p.n + pid[g];
        constexpr int NPRE = PRE ? NP * NT1 * 4 : 0, NX = NP * KP, NL = NPRE + NX + 3 + NP;
        constexpr int PER = (NL + U2 - 1) / U2;
        auto issue = [&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (j < NPRE) {
                constexpr int g = j / (NT1 * 4), t = (j / 4) % NT1, i = j % 4;
                pr[PRE ? g : 0][t][i] = *reinterpret_cast<const float4 *>(p.pre + qn[g] * p.ldp + 32 * t + 8 * i + 4 * h);
            } else if constexpr (j < NPRE + NX) {
                constexpr int g = (j - NPRE) / KP, k = (j - NPRE) % KP;
                float v = 0.f;
                if (k < 3) v = p.xyz[qn[g] * 3 + k];
                else if (k < 3 + p.cf) v = p.feat[qn[g] * p.cf + k - 3];
                xr[g][k] = v;
            } else if constexpr (j < NPRE + NX + 3) {
                constexpr int k = j - NPRE - NX;
                cr[k] = p.new_xyz[(size_t)balln * 3 + k];
            } else {
                constexpr int g = j - NPRE - NX - 3;
                pid[g] = p.idx[(size_t)cnn * (32 * NP) + 32 * g + pt];
            }
        };
        auto hook = [&](auto I) {
#ifndef FPN_SAP_NOGATHER      // (timing-only builds, wrong results: FPN_SAP_NO*)
            constexpr int S = decltype(I)::value;
            static_for<S * PER, ((S + 1) * PER < NL ? (S + 1) * PER : NL)>(issue);
#endif
        };
        TileConst c2[2];        // the second layer's first shifts: wanted after the first layer
        load_const(c2[0], lsh2, 4 * h);
        load_const(c2[1], lsh2, 32 + 4 * h);
        __builtin_amdgcn_sched_barrier(0);
        FPN_PSTAMP(1);

        Act a1[NP][NT1], a2[NP][NT2];
#ifndef FPN_SAP_NOFIRST
        if constexpr (PRE) first.apply(a1, x, h, init);
        else first.apply(a1, x, h);
#else
#pragma unroll
        for (int g = 0; g < NP; ++g)
#pragma unroll
            for (int t = 0; t < NT1; ++t)
#pragma unroll
                for (int f = 0; f < 2; ++f) asm volatile("" : "=v"(a1[g][t].m[f]) : "v"(x[g][0]));
#endif
        FPN_PSTAMP(2);
        layer_t_from<NP, NT1, NT2, 0, PER>(a2, a1, ws, lsh2, feed, h, c2, hook);
        FPN_PSTAMP(3);
        layer_max<NP, NT2, NT3, U2 % kRing>(a2, ws + U2 * kUnitVecs, lsh3, lmax, feed, lane, p.c3);
        FPN_PSTAMP(4);

#ifdef FPN_SAP_NOFLUSH
        if (c + 1 == c1) {
#else
        if (++in_ball == cpb) {             // the ball is complete: its row leaves, the LDS row starts again from zero
#endif
            lds_order();
            float *out = p.out + (size_t)ball * p.ldo;
#pragma unroll
            for (int ch = lane; ch < C; ch += 64) {
                const int v = lmax[ch];
                lmax[ch] = 0;
                if (ch < p.c3) out[ch] = __int_as_float(v);
            }
            lds_order();
            in_ball = 0;
            ++ball;
        }
        FPN_PSTAMP(5);
    }
#ifdef FPN_SA_STAMPS
    if (blockIdx.x == gridDim.x / 2 + 8 && threadIdx.x == 64) {
        long long t = 0;
        reinterpret_cast<long long *>(p.out)[0] = 0;
        for (int i = 1; i < 6; ++i) {
            t += phase[i] / (c1 - c0);
            reinterpret_cast<long long *>(p.out)[i] = t;
        }
    }
#endif
}

// ---------------------------------------------------------------------------
// Per-point head after a feature-propagation step (fp_head_reg_kernel of chain_reg.hip):
// [interpolated features (32 NJA) | own features (cb <= 32)] -> 128 -> 128 -> 128 -> 2.
// Stream: layer 1 ((NJA+1)*32 rows: [xa | xb | zeros]), layer 2, layer 3; the inputs are
// converted to bf16 on the way in, the last hidden layer stays fp32 for the final product.
// ---------------------------------------------------------------------------
struct FpHeadParams {
    long rows;
    int cb;
    int xa_bf16;            // xa holds bf16 values (fpn_three_nn_interpolate_bf16), else float
    const float *xa, *xb;
    const bf16x8 *ws;
    const float *sh1, *sh2, *sh3;
    const float *w4, *b4;
    float *logits;
};

template <int NJA, int NP, class Feed = RingFeed>
__global__ __launch_bounds__(64 * kWaves) void fp_head_bf16_kernel(const FpHeadParams p)
{
    constexpr int NJ = NJA + 1;
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const long row0 = ((long)blockIdx.x * kWaves + (threadIdx.x >> 6)) * (32 * NP);
    if (row0 >= p.rows) return;

    Feed feed;
    feed.start(p.ws, lane, feed_ring<Feed>());
    Act in[NP][NJ], a1[NP][4], a2[NP][4];
    floatx16 a3[NP][4];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        if (p.xa_bf16) {        // (workgroup-uniform) the rows are bf16 already: half the bytes, the same values
            const bf16x4 *xa = reinterpret_cast<const bf16x4 *>(reinterpret_cast<const __bf16 *>(p.xa) +
                                                                 (row0 + 32 * g + pt) * (32 * NJA));
#pragma unroll
            for (int t = 0; t < NJA; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x4 v = xa[8 * t + 2 * q + h];
#pragma unroll
                    for (int e = 0; e < 4; ++e) in[g][t].m[q >> 1][4 * (q & 1) + e] = v[e];
                }
        } else {
            const float *xa = p.xa + (row0 + 32 * g + pt) * (32 * NJA);
#pragma unroll
            for (int t = 0; t < NJA; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(xa + 32 * t + 8 * q + 4 * h);
                    in[g][t].m[q >> 1][4 * (q & 1) + 0] = (__bf16)v.x;
                    in[g][t].m[q >> 1][4 * (q & 1) + 1] = (__bf16)v.y;
                    in[g][t].m[q >> 1][4 * (q & 1) + 2] = (__bf16)v.z;
                    in[g][t].m[q >> 1][4 * (q & 1) + 3] = (__bf16)v.w;
                }
        }
        const float *xb = p.xb + (row0 + 32 * g + pt) * (long)p.cb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 8 * (r >> 2) + 4 * h + (r & 3);
            in[g][NJA].m[r >> 3][r & 7] = (__bf16)(ch < p.cb ? xb[ch] : 0.f);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int U1 = 2 * NJ, U2 = 2 * 4;
    const bf16x8 *ws = p.ws;
    TileConst cs[2];        // (the arithmetic of the persistent kernel, bit for bit)
    load_const(cs[0], p.sh1, 4 * h);
    load_const(cs[1], p.sh1, 32 + 4 * h);
    layer_t_from<NP, NJ, 4, 0>(a1, in, ws, p.sh1, feed, h, cs);
    ws += U1 * kUnitVecs;
    load_const(cs[0], p.sh2, 4 * h);
    load_const(cs[1], p.sh2, 32 + 4 * h);
    layer_t_from<NP, 4, 4, U1 % kRing>(a2, a1, ws, p.sh2, feed, h, cs);
    ws += U2 * kUnitVecs;
    load_const(cs[0], p.sh3, 4 * h);
    load_const(cs[1], p.sh3, 32 + 4 * h);
    layer_t_from<NP, 4, 4, (U1 + U2) % kRing>(a3, a2, ws, p.sh3, feed, h, cs);
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 *w = reinterpret_cast<const float4 *>(p.w4 + (32 * j + 8 * q + 4 * h) * 2);
                const float4 wa = w[0], wb = w[1];
                s0 = fmaf(a3[g][j][4 * q + 0], wa.x, s0); s1 = fmaf(a3[g][j][4 * q + 0], wa.y, s1);
                s0 = fmaf(a3[g][j][4 * q + 1], wa.z, s0); s1 = fmaf(a3[g][j][4 * q + 1], wa.w, s1);
                s0 = fmaf(a3[g][j][4 * q + 2], wb.x, s0); s1 = fmaf(a3[g][j][4 * q + 2], wb.y, s1);
                s0 = fmaf(a3[g][j][4 * q + 3], wb.z, s0); s1 = fmaf(a3[g][j][4 * q + 3], wb.w, s1);
            }
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (h == 0)
            *reinterpret_cast<float2 *>(p.logits + (row0 + 32 * g + pt) * 2) = make_float2(s0 + p.b4[0], s1 + p.b4[1]);
    }
}

// The same head as a PERSISTENT kernel (as sa_first_bf16_kernel): its stream (26 units) and
// constants lie in LDS, a wave walks consecutive groups of 32 NP points and reads the next
// group's inputs - as they lie in memory - while it multiplies this one's.  Launched per
// group, a wave spent three quarters of its time before and after its MFMAs (first weight
// units, inputs, the last layer's weights: each a trip to memory).
template <int NJA, int NP>
__global__ __launch_bounds__(64 * kWaves, 1) void fp_head_lds_bf16_kernel(const FpHeadParams p, unsigned rounds,
                                                                          unsigned rounds_per_wave)
{
    constexpr int NJ = NJA + 1, U1 = 2 * NJ, U2 = 8, U = U1 + 2 * U2;
    static_assert(U % kLdsAhead == 0, "unsupported stack");
    extern __shared__ bf16x8 fp_head_lds[];
    bf16x8 *wl = fp_head_lds;                                           // U + kLdsAhead units
    float *lsh = reinterpret_cast<float *>(wl + (U + kLdsAhead) * kUnitVecs);   // shifts of the three layers
    float *lw4 = lsh + 3 * 128;                                         // (128, 2)
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < U * kUnitVecs; i += 64 * kWaves) wl[i] = p.ws[i];
    for (int i = threadIdx.x; i < kLdsAhead * kUnitVecs; i += 64 * kWaves) wl[U * kUnitVecs + i] = p.ws[i];
    for (int i = threadIdx.x; i < 128; i += 64 * kWaves) {
        lsh[i] = p.sh1[i];
        lsh[128 + i] = p.sh2[i];
        lsh[256 + i] = p.sh3[i];
        lw4[i] = p.w4[i];
        lw4[128 + i] = p.w4[128 + i];
    }
    __syncthreads();
    const unsigned gw = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves + (threadIdx.x >> 6));
    const unsigned r0 = gw * rounds_per_wave, r1 = min(rounds, r0 + rounds_per_wave);
    if (r0 >= r1) return;
    const float b40 = p.b4[0], b41 = p.b4[1];

    // a group's inputs as they lie in memory: 16 x 8 bytes of its interpolated row (bf16) per
    // lane, and the lane's part of the (at most 8) own channels
    constexpr int NQ = 4 * NJA;
    bf16x4 ra[NP][NQ];
    float rb[NP][4];
    auto load_in = [&](unsigned r) {
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const size_t row = (size_t)r * (32 * NP) + 32 * g + pt;
            const bf16x4 *xa = reinterpret_cast<const bf16x4 *>(reinterpret_cast<const __bf16 *>(p.xa) + row * (32 * NJA));
#pragma unroll
            for (int i = 0; i < NQ; ++i) ra[g][i] = xa[2 * i + h];
            if (p.cb > 0) {                 // (uniform)
                const float *xb = p.xb + row * (size_t)p.cb;
#pragma unroll
                for (int i = 0; i < 4; ++i) rb[g][i] = xb[min(4 * h + i, p.cb - 1)];
            }
        }
    };
    load_in(r0);

    LdsFeed feed;
    feed.start(p.ws, lane, wl);
    const bf16x8 *ws = p.ws;
#pragma unroll 1
    for (unsigned r = r0; r < r1; ++r) {
        Act in[NP][NJ];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int t = i >> 2, q = i & 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) in[g][t].m[q >> 1][4 * (q & 1) + e] = ra[g][i][e];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)        // channels 4 h + k of the last tile, k < 4; the rest is padding
                in[g][NJA].m[k >> 3][k & 7] = (__bf16)((k < 4 && 4 * h + k < p.cb) ? rb[g][k & 3] : 0.f);
        }
        load_in(min(r + 1, r1 - 1));
        TileConst cs[2], cn[2];
        load_const(cs[0], lsh, 4 * h);
        load_const(cs[1], lsh, 32 + 4 * h);
        load_const(cn[0], lsh + 128, 4 * h);
        load_const(cn[1], lsh + 128, 32 + 4 * h);
        __builtin_amdgcn_sched_barrier(0);

        Act a1[NP][4], a2[NP][4];
        floatx16 a3[NP][4];
        layer_t_from<NP, NJ, 4, 0>(a1, in, ws, lsh, feed, h, cs);
        load_const(cs[0], lsh + 256, 4 * h);
        load_const(cs[1], lsh + 256, 32 + 4 * h);
        layer_t_from<NP, 4, 4, U1 % kRing>(a2, a1, ws + U1 * kUnitVecs, lsh + 128, feed, h, cn);
        layer_t_from<NP, 4, 4, (U1 + U2) % kRing>(a3, a2, ws + (U1 + U2) * kUnitVecs, lsh + 256, feed, h, cs);
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 *w = reinterpret_cast<const float4 *>(lw4 + (32 * j + 8 * q + 4 * h) * 2);
                    const float4 wa = w[0], wb = w[1];
                    s0 = fmaf(a3[g][j][4 * q + 0], wa.x, s0); s1 = fmaf(a3[g][j][4 * q + 0], wa.y, s1);
                    s0 = fmaf(a3[g][j][4 * q + 1], wa.z, s0); s1 = fmaf(a3[g][j][4 * q + 1], wa.w, s1);
                    s0 = fmaf(a3[g][j][4 * q + 2], wb.x, s0); s1 = fmaf(a3[g][j][4 * q + 2], wb.y, s1);
                    s0 = fmaf(a3[g][j][4 * q + 3], wb.z, s0); s1 = fmaf(a3[g][j][4 * q + 3], wb.w, s1);
                }
            s0 += __shfl_xor(s0, 32);
            s1 += __shfl_xor(s1, 32);
            if (h == 0)
                *reinterpret_cast<float2 *>(p.logits + ((size_t)r * (32 * NP) + 32 * g + pt) * 2) = make_float2(s0 + b40, s1 + b41);
        }
    }
}

// The head with the interpolated features' share of its first layer taken out (as
// fp_head_pre_kernel of chain_reg.hip): the known points' features meet their rows of W1 once,
// the PRODUCTS are interpolated (pre, bf16 rows, scaled like the layer) and start the first
// layer here; the point's own channels (at most 8) add their part through the split-operand
// first layer; then the two remaining 128 x 128 layers (stream: 16 units, in LDS) and the
// final 128 -> 2 product.  Persistent, as fp_head_lds_bf16_kernel.
struct FpHeadPreParams {
    long rows;
    int cb;
    const __bf16 *pre;                  // (rows,128)
    const float *xb;                    // (rows,cb)
    const float *w1b, *sc1, *sh1;       // (cb,128) fp32: the own channels' rows of W1
    const bf16x8 *ws;                   // layer 2, layer 3
    const float *sh2, *sh3, *w4, *b4;
    float *logits;
};

template <int NP>
__global__ __launch_bounds__(64 * kWaves, 1) void fp_head_pre_bf16_kernel(const FpHeadPreParams p, unsigned rounds,
                                                                          unsigned rounds_per_wave)
{
    constexpr int U2 = 8, U = 2 * U2;
    extern __shared__ bf16x8 fp_head_pre_lds[];
    bf16x8 *wl = fp_head_pre_lds;                                       // U + kLdsAhead units
    float *lsh = reinterpret_cast<float *>(wl + (U + kLdsAhead) * kUnitVecs);   // shifts of layers 2 and 3
    float *lw4 = lsh + 2 * 128;                                         // (128, 2)
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < U * kUnitVecs; i += 64 * kWaves) wl[i] = p.ws[i];
    for (int i = threadIdx.x; i < kLdsAhead * kUnitVecs; i += 64 * kWaves) wl[U * kUnitVecs + i] = p.ws[i];
    for (int i = threadIdx.x; i < 128; i += 64 * kWaves) {
        lsh[i] = p.sh2[i];
        lsh[128 + i] = p.sh3[i];
        lw4[i] = p.w4[i];
        lw4[128 + i] = p.w4[128 + i];
    }
    __syncthreads();
    const unsigned gw = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves + (threadIdx.x >> 6));
    const unsigned r0 = gw * rounds_per_wave, r1 = min(rounds, r0 + rounds_per_wave);
    if (r0 >= r1) return;
    const float b40 = p.b4[0], b41 = p.b4[1];
    FirstLayer<4, 8> first;
    first.load(p.w1b, p.cb, p.sc1, p.sh1, lane, h);

    // A group's 32 rows of products are 8 KB in one piece.  The wave copies them into its own LDS image with
    // loads that write LDS directly (1 KB per instruction: lane l of load i lands at byte 1024 i + 16 l, no
    // registers, no staging instructions) and reads its row's 8-byte pieces from there.  A lane fetches whole rows'
    // 16-byte pieces - row 4 i + (l >> 4), piece (l & 15) ^ (row & 15): the XOR on the SOURCE side spreads a
    // column of the image over the banks while the image itself stays lane-linear, as the instruction requires.
    // (Each lane reading its own row straight into registers made every load touch 32 cache lines: the
    // texture path, not HBM, set the pace - 0.26 ms for 2 M points against 0.21 with whole lines.)
    char *stage = reinterpret_cast<char *>(lw4 + 256) + (size_t)(threadIdx.x >> 6) * (NP * 8192);
    float rx[NP][8];
    auto load_in = [&](unsigned r) {
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const size_t row0 = (size_t)r * (32 * NP) + 32 * g;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 4 * i + (int)(lane >> 4);
                const __bf16 *src = p.pre + (row0 + row) * 128 + (((int)(lane & 15) ^ (row & 15)) << 3);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),
                                                 (__attribute__((address_space(3))) void *)(stage + g * 8192 + i * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) rx[g][k] = k < p.cb ? p.xb[(row0 + pt) * p.cb + k] : 0.f;
        }
    };
    load_in(r0);

    LdsFeed feed;
    feed.start(p.ws, lane, wl);
    const bf16x8 *ws = p.ws;
#pragma unroll 1
    for (unsigned r = r0; r < r1; ++r) {
        float x[NP][8];
        floatx16 init[NP][4];
        // ... and the other direction: this round's image was written by loads issued a round ago; they
        // count in vmcnt like every load, and all of them (the image and the point's own channels) are
        // wanted here
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0); lgkmcnt and expcnt untouched
        lds_order();
#pragma unroll
        for (int g = 0; g < NP; ++g) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x[g][k] = rx[g][k];
#pragma unroll
            for (int i = 0; i < 16; ++i) {      // vector 2 i + h of the row = channels 32 (i >> 2) + 8 (i & 3) + 4 h ..
                const bf16x4 v = *reinterpret_cast<const bf16x4 *>(stage + g * 8192 + pt * 256 + ((i ^ (pt & 15)) << 4) + 8 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) init[g][i >> 2][4 * (i & 3) + e] = (float)v[e];
            }
        }
        // The next round's loads write the image through the memory pipe (LDS-direct loads), which the
        // hardware does not order against this wave's outstanding ds_reads of the same bytes: the reads
        // above must have RETURNED (lgkmcnt 0), not merely been issued, before the first such load goes
        // out.  Their values are converted right below, so the wait costs nothing.
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0); vmcnt and expcnt untouched
        lds_order();
        load_in(min(r + 1, r1 - 1));
        TileConst c2[2], c3[2];
        load_const(c2[0], lsh, 4 * h);
        load_const(c2[1], lsh, 32 + 4 * h);
        __builtin_amdgcn_sched_barrier(0);

        Act a1[NP][4], a2[NP][4];
        floatx16 a3[NP][4];
        first.apply(a1, x, h, init);
        load_const(c3[0], lsh + 128, 4 * h);
        load_const(c3[1], lsh + 128, 32 + 4 * h);
        layer_t_from<NP, 4, 4, 0>(a2, a1, ws, lsh, feed, h, c2);
        layer_t_from<NP, 4, 4, U2 % kRing>(a3, a2, ws + U2 * kUnitVecs, lsh + 128, feed, h, c3);
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 *w = reinterpret_cast<const float4 *>(lw4 + (32 * j + 8 * q + 4 * h) * 2);
                    const float4 wa = w[0], wb = w[1];
                    s0 = fmaf(a3[g][j][4 * q + 0], wa.x, s0); s1 = fmaf(a3[g][j][4 * q + 0], wa.y, s1);
                    s0 = fmaf(a3[g][j][4 * q + 1], wa.z, s0); s1 = fmaf(a3[g][j][4 * q + 1], wa.w, s1);
                    s0 = fmaf(a3[g][j][4 * q + 2], wb.x, s0); s1 = fmaf(a3[g][j][4 * q + 2], wb.y, s1);
                    s0 = fmaf(a3[g][j][4 * q + 3], wb.z, s0); s1 = fmaf(a3[g][j][4 * q + 3], wb.w, s1);
                }
            s0 += __shfl_xor(s0, 32);
            s1 += __shfl_xor(s1, 32);
            if (h == 0)
                *reinterpret_cast<float2 *>(p.logits + ((size_t)r * (32 * NP) + 32 * g + pt) * 2) = make_float2(s0 + b40, s1 + b41);
        }
    }
}

// Point features (points, cf) fp32, cf a multiple of 32 -> the operand fragments a lane of the
// set-abstraction kernel holds for that point: vector ((2 t + s) * 2 + h) of a point = the 8
// channels 32 t + 16 s + 8 (e >> 2) + 4 h + (e & 3), e = 0..7, rounded to bf16.
__global__ void pack_features_kernel(long vecs, int cf, const float *__restrict__ feat, bf16x8 *__restrict__ dst)
{
    const int per = cf / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (long)gridDim.x * blockDim.x) {
        const long pnt = i / per;
        const int v = (int)(i % per), h = v & 1, s = (v >> 1) & 1, t = v >> 2;
        const float *src = feat + pnt * cf + 32 * t + 16 * s + 4 * h;
        const float4 a = *reinterpret_cast<const float4 *>(src), c = *reinterpret_cast<const float4 *>(src + 8);
        bf16x8 o;
        o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
        o[4] = (__bf16)c.x; o[5] = (__bf16)c.y; o[6] = (__bf16)c.z; o[7] = (__bf16)c.w;
        dst[i] = o;
    }
}

// w (cin,cout) fp32 row-major, scale (cout) or NULL -> the layer's units:
// element e of fragment (unit (tp, j), tile u of the pair, k-fragment s, lane (h, c)) =
//   bf16( w[32 j + 16 s + 8 (e >> 2) + 4 h + (e & 3)][32 (2 tp + u) + c] * scale[col] ),
// units in the order tp * (cin/32) + j.
__global__ void pack_units_kernel(int cin, int cout, const float *__restrict__ w, const float *__restrict__ scale,
                                  __bf16 *__restrict__ dst)
{
    const long total = (long)cin * cout;
    const int nj = cin / 32;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = i & 7, l = (i >> 3) & 63, s = (i >> 9) & 1, u = (i >> 10) & 1;
        const long unit = i >> 11;
        const int tp = (int)(unit / nj), j = (int)(unit % nj);
        const int k = 32 * j + 16 * s + 8 * (e >> 2) + 4 * (l >> 5) + (e & 3), col = 32 * (2 * tp + u) + (l & 31);
        const float v = w[(long)k * cout + col] * (scale ? scale[col] : 1.f);
        dst[i] = (__bf16)v;
    }
}

// Flipout: the units of one layer for every frustum copy f,
//   W_eff[f] = W_mean + diag(sign_in[f]) . dW[f / frustums_per_delta] . diag(sign_out[f]),
// converted to bf16 in the element order of pack_units_kernel.  One thread writes one
// 16-byte vector per copy for kFlipCopies consecutive copies (blockIdx.y): mean and
// perturbation are read once per group, only the signs differ from copy to copy.
// (4.9 GB a pass of 4096 copies, 1 KB a wave and copy; a plain fill writes 6.9 TB/s on this chip.  What kept the
// kernel at 3.6 TB/s was not its stores but where its float signs come from - see SignBits.)
constexpr int kFlipCopies = 8;

// BITS: the signs are one bit each (set = +1) in the words of `bits`: copy f's input sign k at bit
// in0 + f * ld_in + k, its output sign at bit out0 + f * ld_out + col (in0, ld_in multiples of 4: a lane's four
// consecutive input signs lie in one word).  The float signs of a pass of 4096 copies are 164 MB that come from
// memory in 64-byte pieces between the kernel's writes - with every copy reading the SAME rows the kernel ran in
// 0.94 instead of 1.28 ms a pass; as bits they are 5 MB and stay in the L2.
struct SignBits {
    const unsigned *bits;
    long in0, out0;
};

template <bool BITS>
__global__ __launch_bounds__(256) void flipout_units_kernel(long copies, int cin, int cout, const float *__restrict__ w_mean,
                                                            const float *__restrict__ w_delta, int frustums_per_delta,
                                                            long delta_stride, const float *__restrict__ sign_in,
                                                            int ld_in, const float *__restrict__ sign_out, int ld_out,
                                                            bf16x8 *__restrict__ dst, long copy_vecs, long copy0, SignBits sb_)
{
    const long f0 = copy0 + (long)blockIdx.y * kFlipCopies;
    const int nf = (int)min((long)kFlipCopies, copies - f0);
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= cin * cout / 8) return;
    const int nj = cin / 32;
    const int l = v & 63, s = (v >> 6) & 1, u = (v >> 7) & 1, unit = v >> 8;
    const int tp = unit / nj, j = unit % nj;
    const int col = 32 * (2 * tp + u) + (l & 31), k0 = 32 * j + 16 * s + 4 * (l >> 5);
    float m[8], d[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = w_mean[(long)(k0 + 8 * (e >> 2) + (e & 3)) * cout + col];
    // (all of the group's signs are requested before its first store: loads behind stores wait for the stores)
    typedef float vec4 __attribute__((ext_vector_type(4)));
    vec4 sa[kFlipCopies], sb[kFlipCopies];
    float so[kFlipCopies];
    if constexpr (BITS) {
        unsigned wa[kFlipCopies], wb[kFlipCopies], wo[kFlipCopies];
#pragma unroll
        for (int c = 0; c < kFlipCopies; ++c) {
            const long f = f0 + (c < nf ? c : nf - 1);
            const long ia = sb_.in0 + f * ld_in + k0, io = sb_.out0 + f * ld_out + col;
            wa[c] = sb_.bits[ia >> 5] >> (ia & 31);
            wb[c] = sb_.bits[(ia + 8) >> 5] >> ((ia + 8) & 31);
            wo[c] = sb_.bits[io >> 5] >> (io & 31);
        }
#pragma unroll
        for (int c = 0; c < kFlipCopies; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sa[c][i] = (wa[c] >> i) & 1u ? 1.f : -1.f;
                sb[c][i] = (wb[c] >> i) & 1u ? 1.f : -1.f;
            }
            so[c] = wo[c] & 1u ? 1.f : -1.f;
        }
    } else {
#pragma unroll
        for (int c = 0; c < kFlipCopies; ++c) {
            const long f = f0 + (c < nf ? c : nf - 1);
            const float *si = sign_in + f * ld_in + k0;
            sa[c] = *reinterpret_cast<const vec4 *>(si);
            sb[c] = *reinterpret_cast<const vec4 *>(si + 8);
            so[c] = sign_out[f * ld_out + col];
        }
    }
    long sample = f0 / frustums_per_delta, bound = (sample + 1) * frustums_per_delta;
    {
        const float *dw = w_delta + sample * delta_stride;
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = dw[(long)(k0 + 8 * (e >> 2) + (e & 3)) * cout + col];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < kFlipCopies; ++c) {
        if (c >= nf) break;
        const long f = f0 + c;
        if (f >= bound) {
            ++sample;
            bound += frustums_per_delta;
            const float *dw = w_delta + sample * delta_stride;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = dw[(long)(k0 + 8 * (e >> 2) + (e & 3)) * cout + col];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaf(d[e], (e < 4 ? sa[c][e] : sb[c][e - 4]) * so[c], m[e]);
        dst[f * copy_vecs + v] = o;
    }
}

// FPN_BF16_FEED=ring keeps every wave on its own weight stream (a measurement switch)
inline bool coop_feed()
{
    const char *e = std::getenv("FPN_BF16_FEED");
    return !(e && std::string(e) == "ring");
}

// The cooperative feed needs every wave of every workgroup alive (rows fill the workgroups)
// and on one stream (a workgroup's rows inside one frustum copy when the copies have their
// own streams).  group: rows per copy.
inline bool use_coop(long rows, int np, long copy_vecs, long group)
{
    const long wg_rows = (long)kWaves * 32 * np;
    return coop_feed() && rows % wg_rows == 0 && (copy_vecs == 0 || group % wg_rows == 0);
}

// the kernels copy their layer constants with 16-byte loads (StagedConsts, load_const)
inline bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

inline dim3 grid_for(long rows, int np)
{
    const long waves = rows / (32 * np);
    return dim3((unsigned)((waves + kWaves - 1) / kWaves));
}

// two point groups per wave when a pooling group (g rows) holds whole waves of 64 rows
inline int groups_per_wave(int g)
{
#ifdef FPN_BF16_NP1
    (void)g;
    return 1;
#else
    return g % 64 == 0 ? 2 : 1;
#endif
}

template <int NJF, int NT1, int NT2, int NT3>
int launch_sa(const SaParams &p, hipStream_t st)
{
    const int np = groups_per_wave(p.ns);
    // (the workgroup's LDS ring, CoopFeed, was measured for the stacks over point features as well: 1.08 -> 1.20 ms
    // for the 352-128-128-256 pair of launches, 0.35 -> 0.39 for 160-128-128-256: with two point groups per wave
    // the weight reads are not what these kernels wait for, and the ring's barriers cost more than they save)
    // (KP: the first layer's padded input channels where it reads coordinates - the same rule as launch_sa_first)
    if constexpr (NJF == 0) {
        if (p.cf <= 1) {
            if (np == 2)
                hipLaunchKernelGGL((sa_chain_bf16_kernel<NJF, NT1, NT2, NT3, 2, 4>), grid_for(p.rows, 2), dim3(64 * kWaves), 0, st, p);
            else
                hipLaunchKernelGGL((sa_chain_bf16_kernel<NJF, NT1, NT2, NT3, 1, 4>), grid_for(p.rows, 1), dim3(64 * kWaves), 0, st, p);
            return check_launch("fpn_sa_chain_bf16");
        }
    }
    if (np == 2)
        hipLaunchKernelGGL((sa_chain_bf16_kernel<NJF, NT1, NT2, NT3, 2>), grid_for(p.rows, 2), dim3(64 * kWaves), 0, st, p);
    else
        hipLaunchKernelGGL((sa_chain_bf16_kernel<NJF, NT1, NT2, NT3, 1>), grid_for(p.rows, 1), dim3(64 * kWaves), 0, st, p);
    return check_launch("fpn_sa_chain_bf16");
}

// FPN_BF16_SA=launch keeps the per-group kernel for the stacks that read coordinates (a measurement switch)
inline bool sa_persistent()
{
    const char *e = std::getenv("FPN_BF16_SA");
    return !(e && std::string(e) == "launch");
}

template <int NT1, int NT2, int NT3>
int launch_sa_first(const SaParams &p, hipStream_t st)
{
    constexpr int U = ((NT2 + 1) / 2) * NT1 + (NT3 / 2) * NT2;
    constexpr size_t lds = (size_t)(U + kLdsAhead) * kUnitVecs * 16 + (32 * NT2 + 32 * NT3) * 4 + kWaves * 32 * NT3 * 4;
    static_assert(lds <= 160 * 1024, "the stream does not fit LDS");
    const int cus = current_device_cus();
    if (cus < 1)
        return fail(FPN_E_LAUNCH, "sa_chain_bf16: cannot read the device's CU count");
    const unsigned balls = (unsigned)(p.rows / p.ns);
    const int np = groups_per_wave(p.ns);
    // (one point group per wave leaves registers for two waves per SIMD: two workgroups per CU; measured for
    // balls of 64 and 128 samples too: the same times as two groups per wave at one wave per SIMD)
    unsigned groups = std::min<unsigned>((unsigned)cus * FPN_SAP_OCC(np), (balls + kWaves - 1) / kWaves);
    if (groups >= 8) groups -= groups % 8;          // whole rounds over the XCDs (logical_block)
    const unsigned per_wave = (balls + groups * kWaves - 1) / (groups * kWaves);
    const dim3 grid(groups), block(64 * kWaves);
    if constexpr (NT1 == 4) {
        // The 128-128-256 stack (it exists with first-layer products only): eight waves of ONE point group each per
        // workgroup instead of four waves of two.  Two groups per wave need 431 registers - the spill-over lives in
        // AGPRs and costs ~100 moves per round - one group needs 237, and two waves per SIMD cover each other's
        // epilogues; the weights are read from LDS twice as often, which LDS has room for (0.484 -> 0.465 ms).
        constexpr size_t lds8 = lds + kWaves * 32 * NT3 * 4;
        const unsigned g8 = std::min<unsigned>((unsigned)cus, (balls + 7) / 8);
        const unsigned gg = g8 >= 8 ? g8 - g8 % 8 : g8;
        const unsigned pw = (balls + gg * 8 - 1) / (gg * 8);
        const void *fn8 = reinterpret_cast<const void *>(&sa_first_bf16_kernel<NT1, NT2, NT3, 1, 4, true, 8>);
        if (!allow_dynamic_lds(fn8, lds8))
            return fail(FPN_E_LAUNCH, "sa_chain_bf16: the device refuses %zu bytes of LDS per workgroup", lds8);
        hipLaunchKernelGGL((sa_first_bf16_kernel<NT1, NT2, NT3, 1, 4, true, 8>), dim3(gg), dim3(512), lds8, st, p, balls, pw);
        return check_launch("fpn_sa_chain_bf16");
    } else
    if (p.pre) {        // (with first-layer products there are no direct feature channels: KP = 4)
        const void *fn = np == 2 ? reinterpret_cast<const void *>(&sa_first_bf16_kernel<NT1, NT2, NT3, 2, 4, true>)
                                 : reinterpret_cast<const void *>(&sa_first_bf16_kernel<NT1, NT2, NT3, 1, 4, true>);
        if (!allow_dynamic_lds(fn, lds))
            return fail(FPN_E_LAUNCH, "sa_chain_bf16: the device refuses %zu bytes of LDS per workgroup", lds);
        if (np == 2)
            hipLaunchKernelGGL((sa_first_bf16_kernel<NT1, NT2, NT3, 2, 4, true>), grid, block, lds, st, p, balls, per_wave);
        else
            hipLaunchKernelGGL((sa_first_bf16_kernel<NT1, NT2, NT3, 1, 4, true>), grid, block, lds, st, p, balls, per_wave);
        return check_launch("fpn_sa_chain_bf16");
    }
    if constexpr (lds > 64 * 1024) return fail(FPN_E_UNSUPPORTED, "sa_chain_bf16: this stack exists with first-layer products only");
    else if (np == 2 && p.cf <= 1)
        hipLaunchKernelGGL((sa_first_bf16_kernel<NT1, NT2, NT3, 2, 4>), grid, block, lds, st, p, balls, per_wave);
    else if (np == 2)
        hipLaunchKernelGGL((sa_first_bf16_kernel<NT1, NT2, NT3, 2, 8>), grid, block, lds, st, p, balls, per_wave);
    else if (p.cf <= 1)
        hipLaunchKernelGGL((sa_first_bf16_kernel<NT1, NT2, NT3, 1, 4>), grid, block, lds, st, p, balls, per_wave);
    else
        hipLaunchKernelGGL((sa_first_bf16_kernel<NT1, NT2, NT3, 1, 8>), grid, block, lds, st, p, balls, per_wave);
    return check_launch("fpn_sa_chain_bf16");
}

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" {

int fpn_chain_pad_units_bf16(void) { return kPadUnits; }

int fpn_chain_pack_bf16(int cin, int cout, const float *w, const float *scale, void *units, void *stream)
{
    FPN_REQUIRE(cin >= 32 && cin % 32 == 0 && cout >= 64 && cout % 64 == 0,
                "chain_pack_bf16: cin (%d) must be a multiple of 32 and cout (%d) of 64", cin, cout);
    FPN_REQUIRE(w && units, "chain_pack_bf16: null pointer");
    const long total = (long)cin * cout;
    hipLaunchKernelGGL(pack_units_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
                       as_stream(stream), cin, cout, w, scale, reinterpret_cast<__bf16 *>(units));
    return check_launch("fpn_chain_pack_bf16");
}

int fpn_flipout_stream_bf16(long copies, int cin, int cout, const float *w_mean, const float *w_delta,
                            int frustums_per_delta, long delta_stride, const float *sign_in, int ld_in,
                            const float *sign_out, int ld_out, void *units, long copy_vecs, void *stream)
{
    FPN_REQUIRE(cin >= 32 && cin % 32 == 0 && cout >= 64 && cout % 64 == 0,
                "flipout_stream_bf16: cin (%d) must be a multiple of 32 and cout (%d) of 64", cin, cout);
    FPN_REQUIRE(copies >= 0 && frustums_per_delta >= 1 && ld_in >= cin && ld_out >= cout &&
                    copy_vecs >= (long)cin * cout / 8,
                "flipout_stream_bf16: bad size or stride");
    if (copies == 0) return FPN_OK;
    FPN_REQUIRE(w_mean && w_delta && sign_in && sign_out && units, "flipout_stream_bf16: null pointer");
    const unsigned gx = (unsigned)((cin * cout / 8 + 255) / 256);
    const long per_launch = 65535L * kFlipCopies;
    for (long c0 = 0; c0 < copies; c0 += per_launch) {
        const unsigned gy = (unsigned)((std::min<long>(per_launch, copies - c0) + kFlipCopies - 1) / kFlipCopies);
        hipLaunchKernelGGL(flipout_units_kernel<false>, dim3(gx, gy), dim3(256), 0, as_stream(stream), copies, cin, cout, w_mean, w_delta,
                           frustums_per_delta, delta_stride, sign_in, ld_in, sign_out, ld_out,
                           reinterpret_cast<bf16x8 *>(units), copy_vecs, c0, SignBits{});
    }
    return check_launch("fpn_flipout_stream_bf16");
}

int fpn_flipout_stream_bits_bf16(long copies, int cin, int cout, const float *w_mean, const float *w_delta,
                                 int frustums_per_delta, long delta_stride, const void *bits, long in_bit0, int ld_in,
                                 long out_bit0, int ld_out, void *units, long copy_vecs, void *stream)
{
    FPN_REQUIRE(cin >= 32 && cin % 32 == 0 && cout >= 64 && cout % 64 == 0,
                "flipout_stream_bits_bf16: cin (%d) must be a multiple of 32 and cout (%d) of 64", cin, cout);
    FPN_REQUIRE(copies >= 0 && frustums_per_delta >= 1 && ld_in >= cin && ld_out >= cout &&
                    copy_vecs >= (long)cin * cout / 8 && in_bit0 >= 0 && out_bit0 >= 0,
                "flipout_stream_bits_bf16: bad size or stride");
    FPN_REQUIRE(in_bit0 % 4 == 0 && ld_in % 4 == 0,
                "flipout_stream_bits_bf16: a copy's input signs must start at a multiple of four bits (bit %ld, %d a copy)", in_bit0, ld_in);
    if (copies == 0) return FPN_OK;
    FPN_REQUIRE(w_mean && w_delta && bits && units, "flipout_stream_bits_bf16: null pointer");
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(bits) & 3) == 0, "flipout_stream_bits_bf16: bits must be 4-byte aligned");
    const unsigned gx = (unsigned)((cin * cout / 8 + 255) / 256);
    const long per_launch = 65535L * kFlipCopies;
    const SignBits sb{static_cast<const unsigned *>(bits), in_bit0, out_bit0};
    for (long c0 = 0; c0 < copies; c0 += per_launch) {
        const unsigned gy = (unsigned)((std::min<long>(per_launch, copies - c0) + kFlipCopies - 1) / kFlipCopies);
        hipLaunchKernelGGL(flipout_units_kernel<true>, dim3(gx, gy), dim3(256), 0, as_stream(stream), copies, cin, cout, w_mean, w_delta,
                           frustums_per_delta, delta_stride, nullptr, ld_in, nullptr, ld_out,
                           reinterpret_cast<bf16x8 *>(units), copy_vecs, c0, sb);
    }
    return check_launch("fpn_flipout_stream_bits_bf16");
}

int fpn_v1_seg_head_bf16(long rows, int n, int cpc, const float *pc, const void *wstream, const float *const *consts,
                         long copy_vecs, void *point_feat, float *global_max, int ldg, void *stream)
{
    FPN_REQUIRE(copy_vecs >= 0, "v1_seg_head_bf16: negative stream stride");
    FPN_REQUIRE(rows >= 0 && n >= 1 && cpc >= 3 && cpc <= 8, "v1_seg_head_bf16: bad size rows=%ld n=%d cpc=%d", rows, n, cpc);
    FPN_REQUIRE(n % 32 == 0 && rows % n == 0, "v1_seg_head_bf16: points per frustum (%d) must be a multiple of 32", n);
    FPN_REQUIRE(ldg >= 1024, "v1_seg_head_bf16: leading dimension of global_max (%d) must be >= 1024", ldg);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(pc && wstream && consts && point_feat && global_max, "v1_seg_head_bf16: null pointer");
    for (int i = 0; i < 7; ++i)
        FPN_REQUIRE(consts[i] && aligned16(consts[i]), "v1_seg_head_bf16: constant %d is null or not 16-byte aligned", i);
    FPN_REQUIRE(aligned16(wstream), "v1_seg_head_bf16: the weight stream must be 16-byte aligned");
    HeadParams p{};
    p.rows = rows; p.n = n; p.cpc = cpc; p.pc = pc;
    p.w1 = consts[0]; p.sc1 = consts[1]; p.sh1 = consts[2];
    p.ws = reinterpret_cast<const bf16x8 *>(wstream);
    p.sh2 = consts[3]; p.sh3 = consts[4]; p.sh4 = consts[5]; p.sh5 = consts[6];
    p.point_feat = reinterpret_cast<bf16x8 *>(point_feat); p.gmax = global_max; p.ldg = ldg;
    p.copy_vecs = copy_vecs;
    // (the cooperative feed was measured here too: 0.43 -> 0.41 ms with one stream per frustum copy, nothing
    // with a shared stream: two point groups per wave already use every fragment twice)
    // FPN_BF16_HEAD: "round" = a wave per round of points (any shape), "frustum" = a workgroup per frustum where
    // the shape allows; default: per frustum once there is a frustum for every CU
    const char *form = std::getenv("FPN_BF16_HEAD");
    const std::string hf = form ? form : "";
    const long frustums = rows / n;
    const bool fits = n % (kWaves * 64) == 0 && (long)n * cpc <= kHeadXin && (reinterpret_cast<uintptr_t>(pc) & 15) == 0;
    const int cus = current_device_cus();
    if (fits && hf != "round" && (hf == "frustum" || (cus > 0 && frustums >= cus))) {
        // (the workgroup's LDS ring as the feed - every unit once through the vector memory path - measured here
        // too: conv5 24.9 k cycles a round against 24.1 k)
        // (four point groups per wave - every fragment used four times - fill the register file: conv5's two
        // accumulator sets 256, its input 128, the ring 64; the compiler spilled inside the loop: 0.48 ms against 0.24)
        // with two frustums or more per CU, one workgroup per CU that walks every cus-th frustum (1.2 % over a
        // workgroup per frustum: no workgroup hand-overs); FPN_BF16_HEAD_GRID = another number of workgroups
        const char *ge = std::getenv("FPN_BF16_HEAD_GRID");
        const long grid = ge && std::atol(ge) > 0 ? std::min(frustums, std::atol(ge)) : (cus > 0 && frustums >= 2 * cus) ? cus : frustums;
        hipLaunchKernelGGL((seg_head_frustum_bf16_kernel<2, WrapRingFeedT<kRing>>), dim3((unsigned)grid), dim3(64 * kWaves), 0, as_stream(stream), p);
        return check_launch("fpn_v1_seg_head_bf16");
    }
    if (groups_per_wave(n) == 2)
        hipLaunchKernelGGL(seg_head_bf16_kernel<2>, grid_for(rows, 2), dim3(64 * kWaves), 0, as_stream(stream), p);
    else
        hipLaunchKernelGGL(seg_head_bf16_kernel<1>, grid_for(rows, 1), dim3(64 * kWaves), 0, as_stream(stream), p);
    return check_launch("fpn_v1_seg_head_bf16");
}

int fpn_v1_seg_tail_bf16(long rows, int n, const void *point_feat, const float *group_shift, const void *wstream,
                         const float *const *consts, long copy_vecs, const int *drop_mask, float *logits, void *stream)
{
    FPN_REQUIRE(copy_vecs >= 0, "v1_seg_tail_bf16: negative stream stride");
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(drop_mask) & 15) == 0, "v1_seg_tail_bf16: drop_mask must be 16-byte aligned");
    FPN_REQUIRE(rows >= 0 && n >= 1, "v1_seg_tail_bf16: bad size");
    FPN_REQUIRE(n % 32 == 0 && rows % n == 0, "v1_seg_tail_bf16: points per frustum (%d) must be a multiple of 32", n);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(point_feat && group_shift && wstream && consts && logits, "v1_seg_tail_bf16: null pointer");
    for (int i = 0; i < 4; ++i)
        FPN_REQUIRE(consts[i] && aligned16(consts[i]), "v1_seg_tail_bf16: constant %d is null or not 16-byte aligned", i);
    FPN_REQUIRE(consts[4], "v1_seg_tail_bf16: constant 4 is null");
    FPN_REQUIRE(aligned16(wstream) && aligned16(group_shift) && aligned16(point_feat),
                "v1_seg_tail_bf16: point_feat, group_shift and the weight stream must be 16-byte aligned");
    TailParams p{};
    p.rows = rows; p.n = n; p.point_feat = reinterpret_cast<const bf16x8 *>(point_feat); p.gshift = group_shift;
    p.ws = reinterpret_cast<const bf16x8 *>(wstream);
    p.sh7 = consts[0]; p.sh8 = consts[1]; p.sh9 = consts[2]; p.w10 = consts[3]; p.b10 = consts[4];
    p.logits = logits;
    p.copy_vecs = copy_vecs;
    p.drop = reinterpret_cast<const unsigned *>(drop_mask);
    // FPN_BF16_TAIL: "layers" = the layer-by-layer kernel (the twin the tests compare with), "fused1" / "fused2" =
    // conv6 interleaved with conv7, one / two point groups per wave, "walk2" = fused2 with a workgroup per frustum
    // (or per CU, walking the frustums; FPN_BF16_TAIL_GRID = another number of workgroups) where frustums hold
    // whole rounds of 256 points; default: walk2 once there is a frustum per CU and all frustums share one stream,
    // else fused2 where a frustum holds whole 64-point groups, else fused1
    const char *form = std::getenv("FPN_BF16_TAIL");
    const std::string f = form ? form : "";
    if (f == "layers") {
        if (use_coop(rows, 1, copy_vecs, n))
            hipLaunchKernelGGL(seg_tail_bf16_kernel<CoopFeed>, grid_for(rows, 1), dim3(64 * kWaves), 0, as_stream(stream), p);
        else
            hipLaunchKernelGGL(seg_tail_bf16_kernel<RingFeed>, grid_for(rows, 1), dim3(64 * kWaves), 0, as_stream(stream), p);
        return check_launch("fpn_v1_seg_tail_bf16");
    }
    const int np = (f == "fused1" || n % 64 != 0) ? 1 : 2;
    const long frustums = rows / n;
    const int cus = current_device_cus();
    // (not by default with a stream per frustum copy: a copy's stream is then read by ONE workgroup, four times in
    // a row and out of a cache that 255 other copies pass through meanwhile, instead of by four workgroups side by
    // side - 2.29 against 1.72 ms for 4096 copies)
    if (np == 2 && n % (kWaves * 64) == 0 && f != "fused2" && (f == "walk2" || (copy_vecs == 0 && cus > 0 && frustums >= cus))) {
        const char *ge = std::getenv("FPN_BF16_TAIL_GRID");
        const long grid = ge && std::atol(ge) > 0 ? std::min(frustums, std::atol(ge)) : (cus > 0 && frustums >= 2 * cus) ? cus : frustums;
        hipLaunchKernelGGL((seg_tail_walk_bf16_kernel<2, 2>), dim3((unsigned)grid), dim3(64 * kWaves), 0, as_stream(stream), p);
        return check_launch("fpn_v1_seg_tail_bf16");
    }
    const bool coop = use_coop(rows, np, copy_vecs, n);
    if (np == 2) {      // two point groups per wave: conv7 in two passes of two pairs (all eight tiles' accumulators
                        // for 64 points are the whole accumulator file: the compiler then spills)
        // (every wave reads the stream itself: with each fragment used twice the vector memory path keeps up, and
        // the workgroup's LDS ring - 80 more registers, a barrier every two units - measured slower here)
        // (two, three or four units in flight: 0.440 / 0.443 / 0.436 ms - the weights' latency is not what is left)
        hipLaunchKernelGGL((seg_tail_fused_bf16_kernel<2, 2, RingFeed>), grid_for(rows, 2), dim3(64 * kWaves), 0, as_stream(stream), p);
    } else {
        if (coop) hipLaunchKernelGGL((seg_tail_fused_bf16_kernel<1, 4, CoopFeed>), grid_for(rows, 1), dim3(64 * kWaves), 0, as_stream(stream), p);
        else hipLaunchKernelGGL((seg_tail_fused_bf16_kernel<1, 4, RingFeed>), grid_for(rows, 1), dim3(64 * kWaves), 0, as_stream(stream), p);
    }
    return check_launch("fpn_v1_seg_tail_bf16");
}

int fpn_v1_obj_chain_bf16(long rows, int m, int box, const float *obj, const float *delta, const void *wstream,
                          const float *const *consts, long copy_vecs, float *feat_max, int ldg, void *stream)
{
    FPN_REQUIRE(copy_vecs >= 0, "v1_obj_chain_bf16: negative stream stride");
    FPN_REQUIRE(rows >= 0 && m >= 1, "v1_obj_chain_bf16: bad size");
    FPN_REQUIRE(m % 32 == 0 && rows % m == 0, "v1_obj_chain_bf16: object points per frustum (%d) must be a multiple of 32", m);
    FPN_REQUIRE(ldg >= (box ? 512 : 256), "v1_obj_chain_bf16: leading dimension of feat_max (%d) too small", ldg);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(obj && wstream && consts && feat_max, "v1_obj_chain_bf16: null pointer");
    const int nc = box ? 6 : 5;
    for (int i = 0; i < nc; ++i)
        FPN_REQUIRE(consts[i] && aligned16(consts[i]), "v1_obj_chain_bf16: constant %d is null or not 16-byte aligned", i);
    FPN_REQUIRE(aligned16(wstream), "v1_obj_chain_bf16: the weight stream must be 16-byte aligned");
    ObjParams p{};
    p.rows = rows; p.m = m; p.obj = obj; p.delta = delta;
    p.w1 = consts[0]; p.sc1 = consts[1]; p.sh1 = consts[2];
    p.ws = reinterpret_cast<const bf16x8 *>(wstream);
    p.sh2 = consts[3]; p.sh3 = consts[4]; p.sh4 = box ? consts[5] : nullptr;
    p.gmax = feat_max; p.ldg = ldg;
    p.copy_vecs = copy_vecs;
    const int np = groups_per_wave(m);
    const dim3 grid = grid_for(rows, np), block(64 * kWaves);
    hipStream_t st = as_stream(stream);
    // FPN_BF16_OBJ: "round" = a wave per round of points (any shape), "walk" = workgroups that walk the frustums
    // (one stream for all frustums, whole rounds of 256 points per frustum; FPN_BF16_OBJ_GRID = their number);
    // default: walk from two frustums per CU on
    const char *form = std::getenv("FPN_BF16_OBJ");
    const std::string of = form ? form : "";
    const long frustums = rows / m;
    const int cus = current_device_cus();
    if (np == 2 && copy_vecs == 0 && m % (kWaves * 64) == 0 && of != "round" && (of == "walk" || (cus > 0 && frustums >= 2 * cus))) {
        const char *ge = std::getenv("FPN_BF16_OBJ_GRID");
        const long g = ge && std::atol(ge) > 0 ? std::min(frustums, std::atol(ge)) : (cus > 0 ? std::min<long>(frustums, cus) : frustums);
        if (box) hipLaunchKernelGGL((obj_chain_walk_bf16_kernel<true, 2>), dim3((unsigned)g), block, 0, st, p);
        else hipLaunchKernelGGL((obj_chain_walk_bf16_kernel<false, 2>), dim3((unsigned)g), block, 0, st, p);
        return check_launch("fpn_v1_obj_chain_bf16");
    }
    if (box && np == 2) hipLaunchKernelGGL((obj_chain_bf16_kernel<true, 2>), grid, block, 0, st, p);
    else if (box) hipLaunchKernelGGL((obj_chain_bf16_kernel<true, 1>), grid, block, 0, st, p);
    else if (np == 2) hipLaunchKernelGGL((obj_chain_bf16_kernel<false, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((obj_chain_bf16_kernel<false, 1>), grid, block, 0, st, p);
    return check_launch("fpn_v1_obj_chain_bf16");
}

static int fp_head_chain_bf16(long rows, int ca, int cb, const void *xa_any, int xa_bf16, const float *xb,
                              const void *wstream, const float *const *consts, float *logits, void *stream)
{
    const float *xa = static_cast<const float *>(xa_any);
    FPN_REQUIRE(rows >= 0 && rows % 32 == 0, "fp_head_chain_bf16: rows (%ld) must be a multiple of 32", rows);
    FPN_REQUIRE(ca == 128, "fp_head_chain_bf16: %d interpolated channels, a kernel exists for 128", ca);
    FPN_REQUIRE(cb >= 0 && cb <= 32, "fp_head_chain_bf16: at most 32 own channels per point (%d)", cb);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(xa && (cb == 0 || xb) && wstream && consts && logits, "fp_head_chain_bf16: null pointer");
    for (int i = 0; i < 5; ++i) FPN_REQUIRE(consts[i], "fp_head_chain_bf16: constant %d is null", i);
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(xa) & 15) == 0, "fp_head_chain_bf16: xa must be 16-byte aligned");
    FpHeadParams p{};
    p.rows = rows; p.cb = cb; p.xa_bf16 = xa_bf16; p.xa = xa; p.xb = cb ? xb : (xa_bf16 ? logits : xa);
    p.ws = reinterpret_cast<const bf16x8 *>(wstream);
    p.sh1 = consts[0]; p.sh2 = consts[1]; p.sh3 = consts[2]; p.w4 = consts[3]; p.b4 = consts[4];
    p.logits = logits;
    const int np = groups_per_wave((int)(rows % 64 == 0 ? 64 : 32));
    if (sa_persistent() && xa_bf16 && cb <= 8 && rows / (32 * np) < (1L << 31)) {
        constexpr int U = 2 * 5 + 8 + 8;
        constexpr size_t lds = (size_t)(U + kLdsAhead) * kUnitVecs * 16 + (3 * 128 + 256) * 4;
        const int cus = current_device_cus();
        if (cus < 1)
            return fail(FPN_E_LAUNCH, "fp_head_chain_bf16: cannot read the device's CU count");
        const unsigned rounds = (unsigned)(rows / (32 * np));
        const unsigned groups = std::min<unsigned>((unsigned)cus, (rounds + kWaves - 1) / kWaves);
        const unsigned per_wave = (rounds + groups * kWaves - 1) / (groups * kWaves);
        const void *fn = np == 2 ? reinterpret_cast<const void *>(&fp_head_lds_bf16_kernel<4, 2>)
                                 : reinterpret_cast<const void *>(&fp_head_lds_bf16_kernel<4, 1>);
        if (!allow_dynamic_lds(fn, lds))
            return fail(FPN_E_LAUNCH, "fp_head_chain_bf16: the device refuses %zu bytes of LDS per workgroup", lds);
        if (np == 2)
            hipLaunchKernelGGL((fp_head_lds_bf16_kernel<4, 2>), dim3(groups), dim3(64 * kWaves), lds, as_stream(stream), p, rounds, per_wave);
        else
            hipLaunchKernelGGL((fp_head_lds_bf16_kernel<4, 1>), dim3(groups), dim3(64 * kWaves), lds, as_stream(stream), p, rounds, per_wave);
        return check_launch("fpn_fp_head_chain_bf16");
    }
    if (np == 2)
        hipLaunchKernelGGL((fp_head_bf16_kernel<4, 2>), grid_for(rows, 2), dim3(64 * kWaves), 0, as_stream(stream), p);
    else
        hipLaunchKernelGGL((fp_head_bf16_kernel<4, 1>), grid_for(rows, 1), dim3(64 * kWaves), 0, as_stream(stream), p);
    return check_launch("fpn_fp_head_chain_bf16");
}

int fpn_fp_head_chain_bf16(long rows, int ca, int cb, const float *xa, const float *xb, const void *wstream,
                           const float *const *consts, float *logits, void *stream)
{
    return fp_head_chain_bf16(rows, ca, cb, xa, 0, xb, wstream, consts, logits, stream);
}

int fpn_fp_head_chain_bf16x(long rows, int ca, int cb, const void *xa_bf16, const float *xb, const void *wstream,
                            const float *const *consts, float *logits, void *stream)
{
    return fp_head_chain_bf16(rows, ca, cb, xa_bf16, 1, xb, wstream, consts, logits, stream);
}

int fpn_fp_head_chain_pre_bf16(long rows, int cb, const void *pre_bf16, const float *xb, const void *wstream,
                               const float *const *consts, float *logits, void *stream)
{
    FPN_REQUIRE(rows >= 0 && rows % 32 == 0 && rows / 32 < (1L << 31), "fp_head_chain_pre_bf16: rows (%ld) must be a multiple of 32", rows);
    FPN_REQUIRE(cb >= 0 && cb <= 8, "fp_head_chain_pre_bf16: at most 8 own channels per point (%d)", cb);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(pre_bf16 && (cb == 0 || xb) && wstream && consts && logits, "fp_head_chain_pre_bf16: null pointer");
    for (int i = 0; i < 7; ++i) FPN_REQUIRE(consts[i], "fp_head_chain_pre_bf16: constant %d is null", i);
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(pre_bf16) & 15) == 0, "fp_head_chain_pre_bf16: pre must be 16-byte aligned");
    FpHeadPreParams p{};
    p.rows = rows; p.cb = cb; p.pre = static_cast<const __bf16 *>(pre_bf16); p.xb = cb ? xb : logits;
    p.w1b = consts[0]; p.sc1 = consts[1]; p.sh1 = consts[2];
    p.ws = reinterpret_cast<const bf16x8 *>(wstream);
    p.sh2 = consts[3]; p.sh3 = consts[4]; p.w4 = consts[5]; p.b4 = consts[6];
    p.logits = logits;
    const int cus = current_device_cus();
    if (cus < 1)
        return fail(FPN_E_LAUNCH, "fp_head_chain_pre_bf16: cannot read the device's CU count");
    const int np = rows % 64 == 0 ? 2 : 1;
    // the weight stream, the constants, and every wave's image of its groups' rows of products
    const size_t lds = (size_t)(16 + kLdsAhead) * kUnitVecs * 16 + (2 * 128 + 256) * 4 + (size_t)kWaves * np * 8192;
    const unsigned rounds = (unsigned)(rows / (32 * np));
    const unsigned groups = std::min<unsigned>((unsigned)cus, (rounds + kWaves - 1) / kWaves);
    const unsigned per_wave = (rounds + groups * kWaves - 1) / (groups * kWaves);
    const void *fn = np == 2 ? reinterpret_cast<const void *>(&fp_head_pre_bf16_kernel<2>)
                             : reinterpret_cast<const void *>(&fp_head_pre_bf16_kernel<1>);
    if (!allow_dynamic_lds(fn, lds))
        return fail(FPN_E_LAUNCH, "fp_head_chain_pre_bf16: the device refuses %zu bytes of LDS per workgroup", lds);
    if (np == 2)
        hipLaunchKernelGGL(fp_head_pre_bf16_kernel<2>, dim3(groups), dim3(64 * kWaves), lds, as_stream(stream), p, rounds, per_wave);
    else
        hipLaunchKernelGGL(fp_head_pre_bf16_kernel<1>, dim3(groups), dim3(64 * kWaves), lds, as_stream(stream), p, rounds, per_wave);
    return check_launch("fpn_fp_head_chain_pre_bf16");
}

int fpn_sa_pack_features_bf16(long points, int cf, const float *feat, void *packed, void *stream)
{
    FPN_REQUIRE(points >= 0 && cf >= 32 && cf % 32 == 0, "sa_pack_features_bf16: cf (%d) must be a positive multiple of 32", cf);
    if (points == 0) return FPN_OK;
    FPN_REQUIRE(feat && packed, "sa_pack_features_bf16: null pointer");
    FPN_REQUIRE(((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(packed)) & 15) == 0,
                "sa_pack_features_bf16: arrays must be 16-byte aligned");
    const long vecs = points * (cf / 8);
    hipLaunchKernelGGL(pack_features_kernel, dim3((unsigned)std::min<long>((vecs + 255) / 256, 1 << 20)), dim3(256), 0,
                       as_stream(stream), vecs, cf, feat, static_cast<bf16x8 *>(packed));
    return check_launch("fpn_sa_pack_features_bf16");
}

int fpn_sa_chain_bf16_supported(int cf, int c1, int c2, int c3)
{
    // two group-all stacks (one ball per frustum: 32 points x 640 / 256 features) exist in bf16 only
    if ((cf == 640 && c1 == 128 && c2 == 256 && c3 == 1024) || (cf == 256 && c1 == 256 && c2 == 256 && c3 == 512)) return 1;
    // the bf16 weight streams pair output tiles: widths are multiples of 64, except the middle layer of the
    // 64-96-128 stack, whose stream holds the layer padded to 128 columns and the last layer with 96 rows
    if (c2 % 64 != 0 && !(cf < 32 && c1 == 64 && c2 == 96 && c3 == 128)) return 0;
    return fpn_sa_chain_supported(cf, c1, c2, c3);
}

int fpn_sa_chain_bf16_overwrites(long rows, int cf, int c1, int c2, int c3)
{
    // (the dispatch of sa_chain_bf16 below: the persistent kernels store a ball's pooled row, the others merge maxima)
    return fpn_sa_chain_bf16_supported(cf, c1, c2, c3) && cf < 32 && sa_persistent() && rows < (1L << 31);
}

static int sa_chain_bf16(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                         const float *new_xyz, const float *feat, const void *feat_packed, const float *pre, int ld_pre,
                         const int *idx,
                         const void *wstream, const float *const *consts, float *out, int ldo, int c3_valid,
                         void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 1 && m >= 1 && ns >= 1 && cf >= 0, "sa_chain_bf16: bad size");
    FPN_REQUIRE(ns % 32 == 0, "sa_chain_bf16: samples per ball (%d) must be a multiple of 32", ns);
    FPN_REQUIRE(c3_valid >= 64 && c3_valid % 64 == 0 && c3_valid <= c3 && ldo >= c3_valid,
                "sa_chain_bf16: bad output width %d (c3 %d, ld %d)", c3_valid, c3, ldo);
    if (!fpn_sa_chain_bf16_supported(cf, c1, c2, c3))
        return fail(FPN_E_UNSUPPORTED, "sa_chain_bf16: no kernel for %d feature channels and layer widths %d,%d,%d", cf, c1, c2, c3);
    if (cf < 32 && c1 == 128 && !pre)
        return fail(FPN_E_UNSUPPORTED, "sa_chain_bf16: the 128-wide stack over coordinates exists with first-layer products only");
    if (b == 0) return FPN_OK;
    FPN_REQUIRE(xyz && new_xyz && idx && wstream && consts && out && (cf == 0 || feat || feat_packed), "sa_chain_bf16: null pointer");
    FPN_REQUIRE(!feat_packed || (cf >= 32 && (reinterpret_cast<uintptr_t>(feat_packed) & 15) == 0),
                "sa_chain_bf16: packed features need cf >= 32 and a 16-byte aligned array");
    const int njf = cf >= 32 ? cf / 32 : 0, t1 = c1 / 32, t2 = c2 / 32;
    SaParams p{};
    p.rows = (long)b * m * ns; p.n = n; p.m = m; p.ns = ns; p.cf = cf;
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat = feat; p.idx = idx;
    p.featp = static_cast<const bf16x8 *>(feat_packed);
    p.pre = pre;
    p.ldp = ld_pre;
    p.ws = reinterpret_cast<const bf16x8 *>(wstream);
    if (njf == 0) {       // consts: w1, scale1, shift1, shift2, shift3
        for (int i = 0; i < 5; ++i) FPN_REQUIRE(consts[i], "sa_chain_bf16: constant %d is null", i);
        p.w1 = consts[0]; p.sc1 = consts[1]; p.sh1 = consts[2]; p.sh2 = consts[3]; p.sh3 = consts[4];
    } else {              // consts: shift1, shift2, shift3
        for (int i = 0; i < 3; ++i) FPN_REQUIRE(consts[i], "sa_chain_bf16: constant %d is null", i);
        p.shs1 = consts[0]; p.sh2 = consts[1]; p.sh3 = consts[2];
    }
    p.out = out; p.ldo = ldo; p.c3 = c3_valid;
    hipStream_t st = as_stream(stream);
    if (njf == 0 && sa_persistent() && p.rows < (1L << 31)) {
        if (t1 == 4) return launch_sa_first<4, 4, 8>(p, st);
        if (t2 == 2) return launch_sa_first<2, 2, 4>(p, st);
        if (t2 == 3) return launch_sa_first<2, 3, 4>(p, st);
        return launch_sa_first<2, 4, 4>(p, st);
    }
    if (njf == 0 && t1 == 4) return launch_sa<0, 4, 4, 8>(p, st);
    if (njf == 0 && t2 == 2) return launch_sa<0, 2, 2, 4>(p, st);
    if (njf == 0 && t2 == 3) return launch_sa<0, 2, 3, 4>(p, st);
    if (njf == 0) return launch_sa<0, 2, 4, 4>(p, st);
    if (njf == 10 && t1 == 2) return launch_sa<10, 2, 2, 4>(p, st);
    if (njf == 10) return launch_sa<10, 4, 4, 8>(p, st);
    if (njf == 20) return launch_sa<20, 4, 8, 32>(p, st);
    if (njf == 8) return launch_sa<8, 8, 8, 16>(p, st);
    return launch_sa<4, 4, 4, 8>(p, st);
}

int fpn_sa_chain_bf16(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                      const float *new_xyz, const float *feat, const int *idx, const void *wstream,
                      const float *const *consts, float *out, int ldo, int c3_valid, void *stream)
{
    return sa_chain_bf16(b, n, m, ns, cf, c1, c2, c3, xyz, new_xyz, feat, nullptr, nullptr, 0, idx, wstream, consts, out, ldo,
                         c3_valid, stream);
}

int fpn_sa_chain_pre_bf16(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                          const float *new_xyz, const float *feat, const float *pre, int ld_pre, const int *idx,
                          const void *wstream, const float *const *consts, float *out, int ldo, int c3_valid, void *stream)
{
    FPN_REQUIRE(cf <= 1, "sa_chain_pre_bf16: at most one direct feature channel beside the first-layer products (cf = %d)", cf);
    FPN_REQUIRE((pre || b == 0) && (reinterpret_cast<uintptr_t>(pre) & 15) == 0 && ld_pre >= c1 && ld_pre % 4 == 0,
                "sa_chain_pre_bf16: null or misaligned products, or a row stride (%d) below c1 or no multiple of 4", ld_pre);
    return sa_chain_bf16(b, n, m, ns, cf, c1, c2, c3, xyz, new_xyz, feat, nullptr, pre, ld_pre, idx, wstream, consts, out, ldo,
                         c3_valid, stream);
}

int fpn_sa_chain_bf16p(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                       const float *new_xyz, const void *feat_packed, const int *idx, const void *wstream,
                       const float *const *consts, float *out, int ldo, int c3_valid, void *stream)
{
    FPN_REQUIRE(cf >= 32, "sa_chain_bf16p: packed features exist for cf >= 32 only (cf = %d)", cf);
    return sa_chain_bf16(b, n, m, ns, cf, c1, c2, c3, xyz, new_xyz, nullptr, feat_packed, nullptr, 0, idx, wstream, consts, out,
                         ldo, c3_valid, stream);
}

}  // extern "C"
