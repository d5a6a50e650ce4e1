// tools/fetch_calib.hip -- what rocprofv3's FETCH_SIZE (and the L2's own counters) report for KNOWN byte counts in the two access shapes of
// this library, so that the traversal roofline can be stated in calibrated bytes:
//   stream  : 16 B per lane, coalesced, once over a buffer much larger than the 256 MiB Infinity Cache (the shape of the queue-ordered
//             path state: ray refills, shade prefetches);
//   gather  : the quad-node visit of pt_bvh.h / GeomTop::quad_load -- a lane reads the eight (or six: `vecs`) 16-byte vectors of ONE random
//             128-byte record, the near / far plane vectors picked by three "sign" bits -- over tables of 16 MB, 64 MB (both fit the
//             Infinity Cache; 16 MB is 4 x one XCD's L2) and 2 GB (HBM).  `dep` = 1 makes the next record depend on the loaded data, as a
//             ray's next node does: the records / s of that variant at the traversal kernels' occupancy is the ceiling of a
//             latency-bound walk.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/fetch_calib tools/fetch_calib.hip
//   tools/bin/fetch_calib stream 2048            -> one JSON line: bytes read (known), ms, GB/s
//   tools/bin/fetch_calib gather 64 8 0 [waves]  -> table MB, vectors per record, dependent?, waves per SIMD (default 5)
// tools/fetch_calib.py runs every case under rocprofv3 --pmc and writes profiles/r04_fetch_calib.json.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct alignas(16) v4 { float x, y, z, w; };

__global__ __launch_bounds__(256) void k_fill(v4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { v4 v; v.x = (float)(i & 1023u); v.y = 1.0f; v.z = 2.0f; v.w = (float)((i * 2654435761u) >> 8); p[i] = v; }
}

__global__ __launch_bounds__(256) void k_stream(const v4 *__restrict__ p, size_t n, float *out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const v4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}

__device__ inline uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int VECS, bool DEP>
__global__ __launch_bounds__(256) void k_gather(const char *__restrict__ table, uint32_t rec_mask, uint32_t iters, float *out) {
    uint32_t h = mix((blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u);
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; ++it) {
        h = mix(h + it * 0x9e3779b9u);
        const uint32_t r = (h >> 3) & rec_mask;
        const char *b = table + (size_t)r * 128u;
        const uint32_t px = (h & 1u) ? 16u : 0u, py = 32u + ((h & 2u) ? 16u : 0u), pz = 64u + ((h & 4u) ? 16u : 0u); // the ray's near-plane offsets (quad_near_x/y/z)
        const v4 xn = *(const v4 *)(b + px), xf = *(const v4 *)(b + (px ^ 16u)), yn = *(const v4 *)(b + py), yf = *(const v4 *)(b + (py ^ 16u)), zn = *(const v4 *)(b + pz), zf = *(const v4 *)(b + (pz ^ 16u));
        float s = xn.x + xf.y + yn.z + yf.w + zn.x + zf.y;
        if (VECS == 8) { const v4 refs = *(const v4 *)(b + 96u), meta = *(const v4 *)(b + 112u); s += refs.x + meta.w; }
        acc += s;
        if (DEP) h ^= __float_as_uint(s); // the next record is known only when this one has arrived
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: fetch_calib stream MB | gather MB [vecs 6|8] [dep 0|1] [waves per SIMD] [iters]\n"); return 2; }
    const std::string mode = argv[1];
    const size_t mb = (size_t)std::atoll(argv[2]);
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t bytes = mb << 20, n = bytes / 16;
    v4 *buf; float *out;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, nullptr, buf, n);
    CHECK(hipDeviceSynchronize());
    // evict: a 512 MiB write to another buffer, so that neither L2 nor the Infinity Cache holds the table from the fill
    { v4 *ev; CHECK(hipMalloc(&ev, (size_t)512 << 20)); hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, nullptr, ev, ((size_t)512 << 20) / 16); CHECK(hipDeviceSynchronize()); CHECK(hipFree(ev)); }
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms = 0.0f;
    if (mode == "stream") {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_stream, dim3(cus * 8), dim3(256), 0, nullptr, buf, n, out);
        CHECK(hipEventRecord(b)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, a, b));
        std::printf("{\"case\": \"stream\", \"kernel\": \"k_stream\", \"table_mb\": %zu, \"bytes_read_known\": %zu, \"ms\": %.4f, \"gbs\": %.1f, \"lines_128B\": %zu}\n", mb, bytes, ms, bytes / (ms * 1e-3) / 1e9, bytes / 128);
    } else {
        const int vecs = argc > 3 ? std::atoi(argv[3]) : 8, dep = argc > 4 ? std::atoi(argv[4]) : 0, waves = argc > 5 ? std::atoi(argv[5]) : 5;
        const uint32_t iters = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 64u;
        size_t nrec = bytes / 128; uint32_t mask = 1; while ((size_t)mask * 2 <= nrec) mask *= 2; mask -= 1; // a power of two of records
        const dim3 grid((unsigned)(cus * waves)); // 256-thread workgroups: 4 waves, one per SIMD; `waves` workgroups per CU = waves per SIMD
        CHECK(hipEventRecord(a));
        if (vecs == 8 && !dep) hipLaunchKernelGGL((k_gather<8, false>), grid, dim3(256), 0, nullptr, (const char *)buf, mask, iters, out);
        else if (vecs == 8) hipLaunchKernelGGL((k_gather<8, true>), grid, dim3(256), 0, nullptr, (const char *)buf, mask, iters, out);
        else if (!dep) hipLaunchKernelGGL((k_gather<6, false>), grid, dim3(256), 0, nullptr, (const char *)buf, mask, iters, out);
        else hipLaunchKernelGGL((k_gather<6, true>), grid, dim3(256), 0, nullptr, (const char *)buf, mask, iters, out);
        CHECK(hipEventRecord(b)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, a, b));
        const double reads = (double)cus * waves * 256.0 * iters; // record visits
        const double nr = (double)mask + 1.0;
        const double uniq = nr * (1.0 - std::exp(-reads / nr)); // expected distinct records touched
        std::printf("{\"case\": \"gather\", \"kernel\": \"k_gather<%d, %s>\", \"table_mb\": %.0f, \"records\": %.0f, \"vectors_per_record\": %d, \"dependent\": %d, \"waves_per_simd\": %d, \"record_visits\": %.0f, "
                    "\"bytes_requested_known\": %.0f, \"lines_touched_per_visit\": 1, \"distinct_records_expected\": %.0f, \"ms\": %.4f, \"record_visits_per_s\": %.4g, \"requested_gbs\": %.1f}\n",
                    vecs, dep ? "true" : "false", nr * 128.0 / 1048576.0, nr, vecs, dep, waves, reads, reads * vecs * 16.0, uniq, ms, reads / (ms * 1e-3), reads * vecs * 16.0 / (ms * 1e-3) / 1e9);
    }
    CHECK(hipFree(buf)); CHECK(hipFree(out));
    return 0;
}
