// Round-trip latency of a flag hand-off between two workgroups, by cache-coherence scope and by placement:
//   sc1 (agent scope: what the persistent LSTM kernels of csrc/demucs.hip / csrc/lstm2.hip use -- valid wherever the two workgroups run)
//   sc0 (L2 scope: valid only when both workgroups sit on the SAME XCD, whose L2 they share)
// Workgroups are dealt round-robin over the 8 XCDs: workgroups 0 and 8 share XCD 0, workgroups 0 and 1 do not (the kernel reads
// XCC_ID and reports it).      hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_pingpong tools/micro/xcd_pingpong.hip && /tmp/xcd_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((address_space(1))) unsigned gu32;

template <int MODE>      // 0: sc1 loads / stores, 1: sc0 loads + plain stores
__device__ __forceinline__ unsigned ld(const unsigned* p) {
    unsigned v;
    if (MODE == 0) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__device__ __forceinline__ void st(unsigned* p, unsigned v) {
    if (MODE == 0) asm volatile("global_store_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ void pingpong(unsigned* flags, int partner, int iters, unsigned* xcc, long long* cycles) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = id & 0xf;
    if (blockIdx.x != 0 && (int)blockIdx.x != partner) return;
    if (threadIdx.x != 0) return;
    unsigned* mine = flags + (blockIdx.x == 0 ? 0 : 64);
    unsigned* theirs = flags + (blockIdx.x == 0 ? 64 : 0);
    const long long t0 = wall_clock64();
    for (int i = 1; i <= iters; ++i) {
        if (blockIdx.x == 0) {
            st<MODE>(mine, (unsigned)i);
            unsigned spins = 0;
            while (ld<MODE>(theirs) != (unsigned)i && ++spins < (1u << 14)) {}
        } else {
            unsigned spins = 0;
            while (ld<MODE>(theirs) != (unsigned)i && ++spins < (1u << 14)) {}
            st<MODE>(mine, (unsigned)i);
        }
    }
    if (blockIdx.x == 0) cycles[0] = wall_clock64() - t0;
}

int main() {
    unsigned *flags, *xcc;
    long long* cyc;
    hipMalloc(&flags, 1024); hipMalloc(&xcc, 64 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);      // kHz
    for (int mode = 0; mode < 2; ++mode)
        for (int partner : {8, 1}) {
            if (mode == 1 && partner == 1) continue;      // sc0 across XCDs is not coherent: it would spin to its bound
            hipMemset(flags, 0, 1024);
            hipDeviceSynchronize();
            if (mode == 0) pingpong<0><<<16, 64>>>(flags, partner, iters, xcc, cyc);
            else pingpong<1><<<16, 64>>>(flags, partner, iters, xcc, cyc);
            hipError_t e = hipDeviceSynchronize();
            unsigned hx[16]; long long hc = 0;
            hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s, workgroups 0 (XCC %u) and %d (XCC %u): %s, %.0f ns per round trip (two hand-offs)\n", mode ? "sc0 loads + plain stores" : "sc1 loads / stores",
                   hx[0], partner, hx[partner], hipGetErrorString(e), (double)hc / iters / (rate / 1e6));
        }
    return 0;
}
