// traffic_cal.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access SHAPES of the blend kernels
// (VERDICT r2 item 7).  MI355X_MICROARCH.md establishes "FETCH_SIZE = half the bytes" only for wide coalesced streaming
// reads; tools/pmc_summary.py applied that x2 to every kernel.  Each kernel below moves a byte count that is known
// exactly (and its distinct 64-byte sectors / 128-byte lines are counted on the host), from tables larger than the
// 256 MiB Infinity Cache:
//   cal_stream_read   16 B / lane coalesced stream                      (the guide's case: expect raw = bytes / 2)
//   cal_gather48      one random 48-byte record (three float4) per lane (what blend_fwd / blend_bwd do per list entry)
//   cal_gather16      one random 16-byte record per lane                (the footprint gather of the two-level binning)
//   cal_stream_write  16 B / lane coalesced stream store               (expect raw = bytes)
//   cal_scatter1      one random 1-byte store per lane                  (the qmask[slot] store of blend_fwd<train>)
//   cal_scatter48     one random 48-byte row (three float4 stores) per lane (the gradient rows of blend_bwd)
// Build: hipcc --offload-arch=gfx950 -O3 -o traffic_cal traffic_cal.hip ; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- ./traffic_cal     (and again with WRITE_SIZE)
// It prints one JSON line with the exact byte / sector / line counts per kernel; tools/micro/traffic_cal.sh joins the two.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unordered_set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
constexpr uint32_t kMul = 0x9E3779B1u;   // odd: i -> (i * kMul) & (n - 1) is a permutation of [0, n) for n a power of two

__global__ __launch_bounds__(256) void cal_stream_read(const float4* __restrict__ src, size_t n16, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(src + i));
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void cal_gather48(const float4* __restrict__ table, uint32_t n_rec_mask, uint32_t m, float* sink) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float4* r = table + 3 * (size_t)((i * kMul) & n_rec_mask);
    const float4 a = r[0], b = r[1], c = r[2];
    if (a.x + b.y + c.z == 12345.678f) *sink = a.x;
}
__global__ __launch_bounds__(256) void cal_gather16(const float4* __restrict__ table, uint32_t n_rec_mask, uint32_t m, float* sink) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float4 a = table[(i * kMul) & n_rec_mask];
    if (a.x + a.w == 12345.678f) *sink = a.x;
}
__global__ __launch_bounds__(256) void cal_stream_write(float4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store((f4v){1.f, 2.f, 3.f, (float)i}, reinterpret_cast<f4v*>(dst + i));
}
__global__ __launch_bounds__(256) void cal_scatter1(uint8_t* __restrict__ dst, uint32_t n_mask, uint32_t m) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    dst[(i * kMul) & n_mask] = (uint8_t)i;
}
__global__ __launch_bounds__(256) void cal_scatter48(float4* __restrict__ table, uint32_t n_rec_mask, uint32_t m) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    float4* r = table + 3 * (size_t)((i * kMul) & n_rec_mask);
    r[0] = make_float4(1.f, 2.f, 3.f, 4.f); r[1] = make_float4(5.f, 6.f, 7.f, 8.f); r[2] = make_float4(9.f, 10.f, 11.f, (float)i);
}

// distinct aligned blocks of `gran` bytes touched by m accesses of `width` bytes at byte offsets stride * perm(i)
static size_t distinct_blocks(uint32_t m, uint32_t mask, size_t stride, size_t width, size_t gran) {
    std::unordered_set<uint64_t> s;
    s.reserve((size_t)m * 2);
    for (uint32_t i = 0; i < m; ++i) {
        const size_t lo = (size_t)((i * kMul) & mask) * stride, hi = lo + width - 1;
        for (size_t b = lo / gran; b <= hi / gran; ++b) s.insert(b);
    }
    return s.size();
}

int main() {
    const uint32_t n_rec = 1u << 23;             // 8 Mi records: 384 MiB of 48-byte records, 128 MiB of 16-byte ones
    const uint32_t n_rec16 = 1u << 25;           // 32 Mi 16-byte records = 512 MiB
    const uint32_t n_bytes = 1u << 29;           // 512 MiB byte array for the 1-byte scatter
    const uint32_t m = 1u << 22;                 // 4 Mi accesses per gather / scatter launch
    const size_t stream_bytes = (size_t)1 << 30; // 1 GiB stream
    float4 *table48, *table16, *stream;
    uint8_t* bytes;
    float* sink;
    CK(hipMalloc(&table48, (size_t)n_rec * 48));
    CK(hipMalloc(&table16, (size_t)n_rec16 * 16));
    CK(hipMalloc(&stream, stream_bytes));
    CK(hipMalloc(&bytes, n_bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(table48, 0, (size_t)n_rec * 48));
    CK(hipMemset(table16, 0, (size_t)n_rec16 * 16));
    CK(hipMemset(stream, 0, stream_bytes));
    CK(hipMemset(bytes, 0, n_bytes));
    CK(hipDeviceSynchronize());
    const unsigned gb = (m + 255) / 256;
    for (int rep = 0; rep < 3; ++rep) {   // every launch is preceded by > 256 MiB of other traffic: nothing is Infinity-Cache resident
        hipLaunchKernelGGL(cal_stream_read, dim3(8192), dim3(256), 0, 0, stream, stream_bytes / 16, sink);
        hipLaunchKernelGGL(cal_gather48, dim3(gb), dim3(256), 0, 0, table48, n_rec - 1, m, sink);
        hipLaunchKernelGGL(cal_stream_read, dim3(8192), dim3(256), 0, 0, stream, stream_bytes / 16, sink);
        hipLaunchKernelGGL(cal_gather16, dim3(gb), dim3(256), 0, 0, table16, n_rec16 - 1, m, sink);
        hipLaunchKernelGGL(cal_stream_write, dim3(8192), dim3(256), 0, 0, stream, stream_bytes / 16);
        hipLaunchKernelGGL(cal_scatter1, dim3(gb), dim3(256), 0, 0, bytes, n_bytes - 1, m);
        hipLaunchKernelGGL(cal_stream_write, dim3(8192), dim3(256), 0, 0, stream, stream_bytes / 16);
        hipLaunchKernelGGL(cal_scatter48, dim3(gb), dim3(256), 0, 0, table48, n_rec - 1, m);
        CK(hipDeviceSynchronize());
    }
    printf("{\"cal_stream_read\": {\"bytes\": %zu}, \"cal_stream_write\": {\"bytes\": %zu}, ", stream_bytes, stream_bytes);
    printf("\"cal_gather48\": {\"accesses\": %u, \"bytes\": %zu, \"sectors32\": %zu, \"sectors64\": %zu, \"lines128\": %zu}, ", m, (size_t)m * 48,
           distinct_blocks(m, n_rec - 1, 48, 48, 32), distinct_blocks(m, n_rec - 1, 48, 48, 64), distinct_blocks(m, n_rec - 1, 48, 48, 128));
    printf("\"cal_scatter48\": {\"accesses\": %u, \"bytes\": %zu, \"sectors32\": %zu, \"sectors64\": %zu, \"lines128\": %zu}, ", m, (size_t)m * 48,
           distinct_blocks(m, n_rec - 1, 48, 48, 32), distinct_blocks(m, n_rec - 1, 48, 48, 64), distinct_blocks(m, n_rec - 1, 48, 48, 128));
    printf("\"cal_gather16\": {\"accesses\": %u, \"bytes\": %zu, \"sectors32\": %zu, \"sectors64\": %zu, \"lines128\": %zu}, ", m, (size_t)m * 16,
           distinct_blocks(m, n_rec16 - 1, 16, 16, 32), distinct_blocks(m, n_rec16 - 1, 16, 16, 64), distinct_blocks(m, n_rec16 - 1, 16, 16, 128));
    printf("\"cal_scatter1\": {\"accesses\": %u, \"bytes\": %zu, \"sectors32\": %zu, \"sectors64\": %zu, \"lines128\": %zu}}\n", m, (size_t)m,
           distinct_blocks(m, n_bytes - 1, 1, 1, 32), distinct_blocks(m, n_bytes - 1, 1, 1, 64), distinct_blocks(m, n_bytes - 1, 1, 1, 128));
    return 0;
}
