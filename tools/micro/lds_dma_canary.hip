// Does LDS-DMA (buffer_load ... lds) of one workgroup disturb the LDS of ANOTHER workgroup on the same CU?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_canary tools/micro/lds_dma_canary.hip && /tmp/lds_dma_canary [dma_lds_kb] [oob_fraction_percent]
// Kernel A (second stream): 512 threads, dma_lds_kb of dynamic LDS, every wave streams 1 KB pieces into its slots of the stages by
// LDS-DMA; a fraction of the issues uses an offset beyond num_records (the zero-fill padding conv_gemm_v3 / dtw / wgrad3 rely on).
// Kernel B (first stream): 256 threads, 8 KB of LDS holding a pattern; it re-reads the pattern for a while and counts mismatches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
#define OOB 0x7ffffff0u

__global__ __launch_bounds__(512, 1) void dma_kernel(const unsigned* __restrict__ src, unsigned nbytes, int stages, int iters, int oob_pct, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int s = 0; s < stages; ++s) {
            unsigned char* base = smem + s * 8192 + wv * 1024;
            const unsigned h = (unsigned)(it * 131 + s * 17 + wv * 7 + blockIdx.x * 3);
            const bool oob = (int)(h % 100u) < oob_pct;
            const unsigned vo = oob ? OOB : (unsigned)(((h * 4096u + lane * 16u) % (nbytes - 16u)) & ~15u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)base, 16, vo, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<unsigned*>(smem)[(threadIdx.x * 4 + it) % (stages * 2048)];
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// Second victim: WRITES, atomics and barriers.  Every iteration each thread stores a fresh word, adds 1.0f to one of four LDS counters
// (ds_add_f32, 64 lanes per counter) and, behind a barrier, checks ANOTHER thread's word and the counters -- a lost store, a lost
// atomic or a barrier that lets a read pass a write would all show.
__global__ __launch_bounds__(256) void rw_canary_kernel(int iters, unsigned* __restrict__ bad_count, unsigned* __restrict__ bad_log) {
    __shared__ unsigned word[256];
    __shared__ float cnt[4];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0.f;
    __syncthreads();
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        word[threadIdx.x] = (unsigned)it * 2654435761u + threadIdx.x;
        atomicAdd(&cnt[threadIdx.x & 3], 1.0f);
        __syncthreads();
        const unsigned j = (threadIdx.x * 37u + 11u) & 255u;
        const unsigned got = word[j], want = (unsigned)it * 2654435761u + j;
        const float c = cnt[threadIdx.x & 3], cw = 64.0f * (float)(it + 1);
        if (got != want || c != cw) {
            ++bad;
            const unsigned slot = atomicAdd(bad_count + 1, 1u);
            if (slot < 64) { bad_log[slot * 4] = got != want ? j : 1000u + (threadIdx.x & 3); bad_log[slot * 4 + 1] = got != want ? got : __float_as_uint(c);
                             bad_log[slot * 4 + 2] = (unsigned)it; bad_log[slot * 4 + 3] = blockIdx.x; }
        }
        __syncthreads();
        if (c != cw && threadIdx.x < 4) cnt[threadIdx.x] = cw;      // repair, count every event once
        __syncthreads();
    }
    if (bad) atomicAdd(bad_count, bad);
}

__global__ __launch_bounds__(256) void canary_kernel(int iters, unsigned* __restrict__ bad_count, unsigned* __restrict__ bad_log) {
    __shared__ unsigned pat[2048];                 // 8 KB
    for (int i = threadIdx.x; i < 2048; i += 256) pat[i] = 0xA5000000u | (unsigned)(i * 2654435761u >> 8);
    __syncthreads();
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < 2048; i += 256) {
            const unsigned want = 0xA5000000u | (unsigned)(i * 2654435761u >> 8);
            const unsigned got = reinterpret_cast<volatile unsigned*>(pat)[i];
            if (got != want) {
                ++bad;
                const unsigned slot = atomicAdd(bad_count + 1, 1u);
                if (slot < 64) { bad_log[slot * 4] = (unsigned)i; bad_log[slot * 4 + 1] = got; bad_log[slot * 4 + 2] = (unsigned)it; bad_log[slot * 4 + 3] = blockIdx.x; }
                reinterpret_cast<volatile unsigned*>(pat)[i] = want;      // repair, count every event once
            }
        }
        __syncthreads();
    }
    if (bad) atomicAdd(bad_count, bad);
}

int main(int argc, char** argv) {
    const int lds_kb = argc > 1 ? atoi(argv[1]) : 96;
    const int oob_pct = argc > 2 ? atoi(argv[2]) : 30;
    const int stages = lds_kb * 1024 / 8192;
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const unsigned nbytes = 64u << 20;
    unsigned *src, *sink, *cnt, *log_;
    hipMalloc(&src, nbytes); hipMemset(src, 0x3c, nbytes);
    hipMalloc(&sink, 64); hipMalloc(&cnt, 8); hipMalloc(&log_, 64 * 16);
    hipMemset(cnt, 0, 8); hipMemset(log_, 0, 64 * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipDeviceSynchronize();
    // order: "dma first" (the DMA workgroups take the low LDS addresses of every CU) or "canary first" (argv[3] = workgroups of the
    // canary kernel resident per CU before the DMA kernel arrives: its 96 KB then start ABOVE their 8 KB each)
    const int first = argc > 3 ? atoi(argv[3]) : 0;
    const int rw = argc > 4 ? atoi(argv[4]) : 0;           // 1: the write / atomic / barrier victim instead of the read-only one
    for (int rep = 0; rep < 20; ++rep) {
        if (first > 0) {
            if (rw) rw_canary_kernel<<<256 * first, 256, 0, s1>>>(20000, cnt, log_);
            else canary_kernel<<<256 * first, 256, 0, s1>>>(3000, cnt, log_);
            hipStreamQuery(s1);
            usleep(300);
            dma_kernel<<<256, 512, (size_t)lds_kb * 1024, s2>>>(src, nbytes, stages, 400, oob_pct, sink);
            hipDeviceSynchronize();
        } else {
            dma_kernel<<<256, 512, (size_t)lds_kb * 1024, s2>>>(src, nbytes, stages, 400, oob_pct, sink);
            if (rw) rw_canary_kernel<<<2048, 256, 0, s1>>>(3000, cnt, log_);
            else canary_kernel<<<2048, 256, 0, s1>>>(300, cnt, log_);
        }
    }
    hipError_t e = hipDeviceSynchronize();
    unsigned h[2];
    std::vector<unsigned> hl(64 * 4);
    hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost);
    hipMemcpy(hl.data(), log_, 64 * 16, hipMemcpyDeviceToHost);
    printf("%s victim, %s: dma lds %d KB (%d stages), oob %d %%: %s; mismatching words seen by the canary workgroups: %u (events %u)\n", rw ? "write/atomic/barrier" : "read-only",
           first ? "canary first" : "dma first", lds_kb, stages, oob_pct,
           hipGetErrorString(e), h[0], h[1]);
    for (unsigned i = 0; i < (h[1] < 16 ? h[1] : 16); ++i)
        printf("   word %u got 0x%08x (iteration %u, block %u)\n", hl[i * 4], hl[i * 4 + 1], hl[i * 4 + 2], hl[i * 4 + 3]);
    return 0;
}
