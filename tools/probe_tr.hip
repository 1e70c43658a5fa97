// probe: what does ds_read_b64_tr_b16 return?  LDS holds u16 value == its own u16 index.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
extern "C" __global__ void probe_tr(const int* lane_byte_addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = lane_byte_addr[threadIdx.x];
    auto p = (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + a);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
extern "C" int run_probe(const int* addr_dev, unsigned short* out_dev, void* stream) {
    hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, (hipStream_t)stream, addr_dev, out_dev);
    return (int)hipGetLastError();
}
