// ta_rate.hip — how fast can an MI355X serve per-lane ("divergent") loads?
//
// The BVH walk of the path tracer issues, per visited node or tested triangle, one 16-byte buffer load per lane at an
// address of the lane's own: a wave-instruction then touches up to 64 different cache lines, and the texture
// addresser / L1 tag pipeline takes them one after the other.  The walk kernels are bound by that rate, not by HBM
// bandwidth and not by arithmetic, so their roofline is "per-lane accesses per second".  This program measures the
// peak: every lane loads from pseudo-random 16-byte-aligned slots of a table of a given footprint (L2-resident for the
// footprints of interest), either independently (many loads in flight: the peak) or as a dependent chain (the next
// slot comes out of the loaded words: what a walk does).  It sweeps load width (4 / 8 / 16 bytes), the number of active
// lanes per wave and the number of resident waves, and prints one JSON object.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/ta_rate tools/ta_rate.hip
//   tools/ta_rate [--footprint BYTES] [--json]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(e)                                                                                         \
    do {                                                                                                 \
        hipError_t e_ = (e);                                                                             \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); std::exit(1); } \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// WIDTH = bytes per lane per load; CHAIN: the next slot depends on the loaded data; SAME_LINE: all lanes of a wave read
// the same 64 consecutive slots (a coalesced wave-instruction, for reference).  `active` lanes of every wave take part.
// NEAR > 0 (chains only): of every NEAR + 1 loads of a lane, NEAR go to the slot after the previous one (the lane's own
// cache line again, unless it ended there) — what a depth-first node order gives a walk.  NEAR < 0: every step loads a
// slot AND the slot after it, both in flight together.
// GROUP > 1: GROUP neighbouring lanes read the GROUP consecutive 16-byte pieces of one random record of 16 * GROUP bytes
// (aligned to its size): what a fetch of a 32- or 64-byte record shared out over a lane pair or quad looks like.
template <int WIDTH, bool CHAIN, bool SAME_LINE, int NEAR = 0, int GROUP = 1>
__global__ __launch_bounds__(256) void k_loads(const uint4 *table, uint32_t n_slots, uint32_t iters, uint32_t active, uint32_t *sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(table), (short)0, (int)(n_slots * 16u), 0x00020000);
    // which lanes take part: a fixed pseudo-random subset of the wave, the same in every wave
    const bool on = (GROUP == 4 && active == 48u) ? (lane % 4u) < 3u // (three lanes of every quad)
                                                   : (mix32(lane * 0x9E3779B9u + 12345u) % 64u) < active || active >= 64u;
    uint32_t acc = 0;
    if (on) {
        uint32_t s = mix32(SAME_LINE ? (gid >> 6) : (GROUP > 1 ? gid / GROUP : gid));
        if (CHAIN && GROUP > 1) {
            uint32_t rec = __umulhi(s, n_slots / GROUP);
            for (uint32_t i = 0; i < iters; i++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((rec * GROUP + lane % GROUP) * 16u), 0, 0);
                uint32_t w = v.x ^ v.w;
                for (int sh = 1; sh < GROUP; sh <<= 1) w ^= (uint32_t)__shfl_xor((int)w, sh); // the group goes on together
                acc += w;
                rec = __umulhi(mix32(w + i + (gid / GROUP) * 0x9E3779B9u), n_slots / GROUP); // (with the group's id: chains of different groups never merge)
            }
        } else if (CHAIN) {
            uint32_t slot = __umulhi(s, n_slots);
            for (uint32_t i = 0; i < iters; i++) {
                uint32_t w;
                if (NEAR < 0) { // the slot and the one after it, both in flight together: one round trip for two loads
                    const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(slot * 16u), 0, 0);
                    const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(slot * 16u + 16u), 0, 0);
                    w = v0.x ^ v0.w ^ v1.y;
                    i++;
                } else
                if (WIDTH == 16) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(slot * 16u), 0, 0); w = v.x ^ v.w; }
                else if (WIDTH == 8) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(slot * 16u), 0, 0); w = v.x ^ v.y; }
                else w = __builtin_amdgcn_raw_buffer_load_b32(r, (int)(slot * 16u), 0, 0);
                acc += w;
                if (NEAR > 0 && (i % (NEAR + 1)) != 0u) slot = slot + 1u < n_slots ? slot + 1u : 0u;
                else slot = __umulhi(mix32(w + i + gid * 0x9E3779B9u), n_slots); // with i and the lane's id: no cycles, and chains of different lanes never merge
            }
        } else {
            for (uint32_t i = 0; i < iters; i += 8) {
                uint32_t w[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    s = s * 1664525u + 1013904223u;
                    uint32_t slot = SAME_LINE ? (__umulhi(mix32(s), n_slots / 64u) * 64u + lane)
                                  : GROUP > 1 ? (__umulhi(mix32(s), n_slots / GROUP) * GROUP + lane % GROUP) : __umulhi(mix32(s), n_slots);
                    if (WIDTH == 16) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(slot * 16u), 0, 0); w[k] = v.x ^ v.w; }
                    else if (WIDTH == 8) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(slot * 16u), 0, 0); w[k] = v.x ^ v.y; }
                    else w[k] = __builtin_amdgcn_raw_buffer_load_b32(r, (int)(slot * 16u), 0, 0);
                }
#pragma unroll
                for (int k = 0; k < 8; k++) acc += w[k];
            }
        }
    }
    if (acc == 0x12345u) sink[gid & 1023u] = acc; // keeps the loads alive
}

// Does it pay to keep every XCD (workgroup id mod 8: the dispatcher deals workgroups out to the eight dies in turn) to
// an eighth of the table of its own?  Each die's L2 then holds 1/8 of the footprint.  Independent 16-byte loads.
template <bool SLICED>
__global__ __launch_bounds__(256) void k_sliced(const uint4 *table, uint32_t n_slots, uint32_t iters, uint32_t *sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(table), (short)0, (int)(n_slots * 16u), 0x00020000);
    const uint32_t per = n_slots / 8u, base = SLICED ? (blockIdx.x % 8u) * per : 0u, span = SLICED ? per : n_slots;
    uint32_t s = mix32(gid), acc = 0;
    for (uint32_t i = 0; i < iters; i += 8) {
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            s = s * 1664525u + 1013904223u;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((base + __umulhi(mix32(s), span)) * 16u), 0, 0);
            w[k] = v.x ^ v.w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += w[k];
    }
    if (acc == 0x12345u) sink[gid & 1023u] = acc;
}

// One lane fetches a whole record by itself: PIECES loads of 16 bytes at consecutive addresses, records STRIDE bytes apart
// (48: packed triangle records, which straddle cache lines; 32 / 64: aligned records).  Independent records, 2 per round.
template <int PIECES, int STRIDE>
__global__ __launch_bounds__(256) void k_records(const uint4 *table, uint32_t n_slots, uint32_t iters, uint32_t *sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(table), (short)0, (int)(n_slots * 16u), 0x00020000);
    const uint32_t n_rec = (n_slots * 16u) / STRIDE - 1u;
    uint32_t s = mix32(gid), acc = 0;
    for (uint32_t i = 0; i < iters; i += 2) {
        uint32_t w[2 * PIECES];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            s = s * 1664525u + 1013904223u;
            const uint32_t off = __umulhi(mix32(s), n_rec) * STRIDE;
#pragma unroll
            for (int p = 0; p < PIECES; p++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + 16u * p), 0, 0);
                w[k * PIECES + p] = v.x ^ v.w;
            }
        }
#pragma unroll
        for (int k = 0; k < 2 * PIECES; k++) acc += w[k];
    }
    if (acc == 0x12345u) sink[gid & 1023u] = acc;
}

struct Result {
    std::string name;
    int width, active, waves_per_simd;
    bool chain, same_line;
    double ms, glane, gwave;
};

template <int WIDTH, bool CHAIN, bool SAME_LINE, int NEAR = 0, int GROUP = 1>
static Result run(const char *name, const uint4 *table, uint32_t n_slots, uint32_t active, int waves_per_simd, uint32_t *sink, int n_cu) {
    const uint32_t blocks = (uint32_t)(n_cu * waves_per_simd); // 4 waves per block, 4 SIMDs per CU
    uint32_t iters = CHAIN ? 2048u : 8192u;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 0.0f;
    for (int rep = 0; rep < 5; rep++) { // the first launch warms the caches; the best of the others counts
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_loads<WIDTH, CHAIN, SAME_LINE, NEAR, GROUP>), dim3(blocks), dim3(256), 0, 0, table, n_slots, iters, active, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (rep == 1 || (rep > 1 && t < ms)) ms = t;
    }
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    const double waves = (double)blocks * 4.0;
    const double lanes_on = active >= 64u ? 64.0 : (double)active; // expected; the subset is pseudo-random
    Result r{name, WIDTH, (int)active, waves_per_simd, CHAIN, SAME_LINE, ms, waves * lanes_on * iters / (ms * 1e-3) / 1e9, waves * iters / (ms * 1e-3) / 1e9};
    return r;
}

template <bool SLICED>
static Result run_sliced(const char *name, const uint4 *table, uint32_t n_slots, uint32_t *sink, int n_cu) {
    const uint32_t blocks = (uint32_t)(n_cu * 8), iters = 8192u;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 0.0f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_sliced<SLICED>), dim3(blocks), dim3(256), 0, 0, table, n_slots, iters, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (rep == 1 || (rep > 1 && t < ms)) ms = t;
    }
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    const double waves = (double)blocks * 4.0;
    Result r{name, 16, 64, 8, false, false, ms, waves * 64.0 * iters / (ms * 1e-3) / 1e9, waves * iters / (ms * 1e-3) / 1e9};
    return r;
}

template <int PIECES, int STRIDE>
static Result run_records(const char *name, const uint4 *table, uint32_t n_slots, int waves_per_simd, uint32_t *sink, int n_cu) {
    const uint32_t blocks = (uint32_t)(n_cu * waves_per_simd);
    const uint32_t iters = 4096u;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 0.0f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_records<PIECES, STRIDE>), dim3(blocks), dim3(256), 0, 0, table, n_slots, iters, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (rep == 1 || (rep > 1 && t < ms)) ms = t;
    }
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    const double waves = (double)blocks * 4.0;
    // Glane_per_s counts the 16-byte loads; records per second = that / PIECES
    Result r{name, 16, 64, waves_per_simd, false, false, ms, waves * 64.0 * iters * PIECES / (ms * 1e-3) / 1e9, waves * iters * PIECES / (ms * 1e-3) / 1e9};
    return r;
}

int main(int argc, char **argv) {
    size_t footprint = 6u << 20; // node arrays + triangle records of the 31 k-triangle mesh scene: about 5-6 MB
    bool json = false;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--footprint") && i + 1 < argc) footprint = (size_t)std::atoll(argv[++i]);
        else if (!std::strcmp(argv[i], "--json")) json = true;
    }
    if (footprint < 4096) footprint = 4096;
    if (footprint > ((size_t)1 << 31)) footprint = (size_t)1 << 31;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const uint32_t n_slots = (uint32_t)(footprint / 16) & ~63u;
    std::vector<uint32_t> host((size_t)n_slots * 4);
    uint32_t s = 1;
    for (auto &w : host) { s = s * 1664525u + 1013904223u; w = mix32(s); }
    uint4 *table;
    uint32_t *sink;
    CHECK(hipMalloc((void **)&table, (size_t)n_slots * 16));
    CHECK(hipMalloc((void **)&sink, 4096));
    CHECK(hipMemcpy(table, host.data(), (size_t)n_slots * 16, hipMemcpyHostToDevice));

    std::vector<Result> rs;
    // the peak: independent 16-byte loads, all lanes, 8 waves per SIMD
    rs.push_back(run<16, false, false>("divergent 16 B, independent", table, n_slots, 64, 8, sink, n_cu));
    rs.push_back(run<8, false, false>("divergent 8 B, independent", table, n_slots, 64, 8, sink, n_cu));
    rs.push_back(run<4, false, false>("divergent 4 B, independent", table, n_slots, 64, 8, sink, n_cu));
    rs.push_back(run<16, false, true>("coalesced 16 B (one KiB per wave-instruction), independent", table, n_slots, 64, 8, sink, n_cu));
    // does a wave-instruction cost by its active lanes?
    for (uint32_t a : {48u, 32u, 16u, 8u}) rs.push_back(run<16, false, false>("divergent 16 B, independent, part of the lanes", table, n_slots, a, 8, sink, n_cu));
    // what a walk does: a dependent chain per lane; the rate then depends on the waves in flight
    for (int w : {2, 4, 6, 8}) rs.push_back(run<16, true, false>("divergent 16 B, dependent chain", table, n_slots, 64, w, sink, n_cu));
    rs.push_back(run<16, true, false>("divergent 16 B, dependent chain, part of the lanes", table, n_slots, 32, 8, sink, n_cu));
    rs.push_back(run<4, true, false>("divergent 4 B, dependent chain", table, n_slots, 64, 8, sink, n_cu));
    // every second / three of four loads in the line of the previous one (L1 hits)
    rs.push_back(run<16, true, false, 1>("divergent 16 B, dependent chain, 1 of 2 loads in the previous line", table, n_slots, 64, 8, sink, n_cu));
    rs.push_back(run<16, true, false, 3>("divergent 16 B, dependent chain, 3 of 4 loads in the previous line", table, n_slots, 64, 8, sink, n_cu));
    for (int w : {2, 4, 8}) rs.push_back(run<16, true, false, -1>("divergent 16 B, dependent chain of load PAIRS (slot, slot + 1)", table, n_slots, 64, w, sink, n_cu));
    // records of 32 / 64 bytes fetched by lane pairs / quads: 32 / 16 different records per wave-instruction
    for (int w : {2, 4, 8}) {
        rs.push_back(run<16, true, false, 0, 2>("records of 32 B by lane pairs (16 B per lane), dependent chain", table, n_slots, 64, w, sink, n_cu));
        rs.push_back(run<16, true, false, 0, 4>("records of 64 B by lane quads (16 B per lane), dependent chain", table, n_slots, 64, w, sink, n_cu));
    }
    rs.push_back(run<16, false, false, 0, 2>("records of 32 B by lane pairs (16 B per lane), independent", table, n_slots, 64, 8, sink, n_cu));
    rs.push_back(run<16, false, false, 0, 4>("records of 64 B by lane quads (16 B per lane), independent", table, n_slots, 64, 8, sink, n_cu));
    rs.push_back(run<16, false, false, 0, 2>("records of 32 B by lane pairs (16 B per lane), independent", table, n_slots, 64, 2, sink, n_cu));
    rs.push_back(run<16, false, false, 0, 4>("records of 64 B by lane quads (16 B per lane), independent", table, n_slots, 64, 2, sink, n_cu));
    rs.push_back(run<16, false, false, 0, 8>("records of 128 B by 8 lanes (16 B per lane), independent", table, n_slots, 64, 8, sink, n_cu));
    rs.push_back(run<16, false, false, 0, 4>("records of 64 B by lane quads, 3 of 4 lanes on, independent", table, n_slots, 48, 8, sink, n_cu));
    rs.push_back(run<16, false, false>("divergent 16 B, independent", table, n_slots, 64, 2, sink, n_cu));
    rs.push_back(run<16, false, false>("divergent 16 B, independent", table, n_slots, 64, 4, sink, n_cu));
    // a lane fetches its record alone, 16 bytes per load
    rs.push_back(run_records<2, 32>("records of 32 B, one lane each (2 loads), independent", table, n_slots, 8, sink, n_cu));
    rs.push_back(run_records<3, 48>("records of 48 B packed, one lane each (3 loads), independent", table, n_slots, 8, sink, n_cu));
    rs.push_back(run_records<3, 64>("records of 48 B in 64 B slots, one lane each (3 loads), independent", table, n_slots, 8, sink, n_cu));
    rs.push_back(run_records<4, 64>("records of 64 B, one lane each (4 loads), independent", table, n_slots, 8, sink, n_cu));
    rs.push_back(run_records<1, 16>("records of 16 B, one lane each (1 load), independent", table, n_slots, 8, sink, n_cu));

    // one eighth of the table per XCD
    rs.push_back(run_sliced<false>("divergent 16 B, independent, every workgroup over the whole table", table, n_slots, sink, n_cu));
    rs.push_back(run_sliced<true>("divergent 16 B, independent, workgroup b over eighth (b mod 8) of the table", table, n_slots, sink, n_cu));

    if (json) {
        std::printf("{\"device\": \"%s\", \"compute_units\": %d, \"footprint_bytes\": %zu, \"peak_divergent_16B_Glane_per_s\": %.2f, \"rows\": [", prop.gcnArchName, n_cu,
                    (size_t)n_slots * 16, rs[0].glane);
        for (size_t i = 0; i < rs.size(); i++)
            std::printf("%s{\"what\": \"%s\", \"bytes\": %d, \"active_lanes\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"Glane_per_s\": %.2f, \"Gwave_instr_per_s\": %.3f}",
                        i ? ", " : "", rs[i].name.c_str(), rs[i].width, rs[i].active, rs[i].waves_per_simd, rs[i].ms, rs[i].glane, rs[i].gwave);
        std::printf("]}\n");
    } else {
        std::printf("%s, %d CUs, table of %zu bytes\n", prop.gcnArchName, n_cu, (size_t)n_slots * 16);
        for (const Result &r : rs)
            std::printf("%-72s %2d B  %2d lanes  %d waves/SIMD  %8.3f ms  %8.2f G lane-loads/s  %7.3f G wave-instr/s\n", r.name.c_str(), r.width, r.active,
                        r.waves_per_simd, r.ms, r.glane, r.gwave);
    }
    CHECK(hipFree(table));
    CHECK(hipFree(sink));
    return 0;
}
