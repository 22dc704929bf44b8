// Fixed-order sums for the deterministic schedule (sehip_set_deterministic; the reference ships `cudnn_deterministic: True`,
// src/conf/config.yaml:130, and applies it whatever the model: src/utils.py:108-111).
//
// The normalisation kernels of ConvTasNet (csrc/tasnet.hip) and Demucs (csrc/demucs.hip) add per-utterance / per-group sums from many
// workgroups into one double each.  Default schedule: one double atomic per workgroup and value -- the order of arrival varies from
// run to run and so does the last bit of the sum.  Deterministic schedule: every workgroup stores its values into ITS slot of a
// partial array and a second, tiny launch (sehip_det_finish) adds the slots in a fixed order into the destination: lane l of one
// wave per (group, value) adds the slots l, l + 64, ... in increasing order, the 64 lane sums meet in a fixed xor tree.
//
// (A first version let the LAST workgroup of the launch add the slots -- ticket counter behind a device-scope release, the hand-off of
//  csrc/rbn.hip's opt-in finalize.  Alone it was bit-stable over hundreds of calls (tools/dev/det_actbwd.py); inside the Demucs step,
//  beside the weight-gradient stream, one value of one launch in ~100 came out different by 5e-4 relative while the same call repeated
//  right behind it on the same operands gave the reference value three times (tools/dev/det_diff.py with SEHIP_DET_DEBUG=1).  Device-
//  scope stores / loads for the slots did not change that.  Not understood; the launch boundary is the hand-off that is certain.)
#pragma once
#include "common.h"

struct DetCtx {
    double* part;       // [group][workgroup of the group][value]; nullptr: the default schedule (atomics)
};

// Host (csrc/api.cpp): the per-stream partial array of the deterministic schedule; {nullptr} when the schedule is off.  ok = false: the
// schedule is on and the array could not be had (allocation failure, inside a stream capture).
DetCtx sehip_det_ctx(hipStream_t st, size_t ndoubles, bool* ok);
// dst[g * dst_stride + i] += sum over b < nb of part[(g * nb + b) * nvals + i], fixed order; no-op when dc.part == nullptr
int sehip_det_finish(hipStream_t st, DetCtx dc, int ngroups, int nb, int nvals, double* dst, long dst_stride);

// Every thread of the workgroup calls this; vals[0 .. nvals) (LDS, float or double) are complete and visible (a __syncthreads()
// lies between their last write and this call).  Default schedule: dst[i] += vals[i] by double atomics.  Deterministic schedule: the
// values go to slot (group, bidx) of the partial array; the caller's sehip_det_finish adds the slots afterwards.
template <typename T>
__device__ __forceinline__ void det_group_add(const T* vals, int nvals, double* dst, const DetCtx dc, int group, int bidx, int nb) {
    if (dc.part == nullptr) {
        for (int i = threadIdx.x; i < nvals; i += blockDim.x) atomicAdd(&dst[i], (double)vals[i]);
        return;
    }
    double* mine = dc.part + ((size_t)group * nb + bidx) * nvals;
    for (int i = threadIdx.x; i < nvals; i += blockDim.x) mine[i] = (double)vals[i];
}

// Inside a workgroup of <= 256 threads: out[s] (+)= sum of the v of the threads whose slot is s, added in thread order by ONE thread
// per slot (the default schedule does this with LDS atomics, whose order is the hardware's).  tmpv / tmps: blockDim.x words of LDS each.
__device__ __forceinline__ void det_slot_sum(float v, int slot, int nslots, float* out, float* tmpv, int* tmps, bool accumulate) {
    tmpv[threadIdx.x] = v;
    tmps[threadIdx.x] = slot;
    __syncthreads();
    if ((int)threadIdx.x < nslots) {
        float a = accumulate ? out[threadIdx.x] : 0.f;
        for (int t = 0; t < (int)blockDim.x; ++t)
            if (tmps[t] == (int)threadIdx.x) a += tmpv[t];
        out[threadIdx.x] = a;
    }
    __syncthreads();
}
