// Hardware probe (MI355X / gfx950): what does `buffer_load_dword[x4] ... offen lds` write to LDS for a lane whose
// offset is beyond the buffer's num_records?  (The stream-K convolution stages padding taps that way.)
//   build: hipcc --offload-arch=gfx950 -O2 buffer_lds_oob.hip -o buffer_lds_oob ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ void probe(const float* x, int n_floats, float* out) {
    __shared__ __attribute__((aligned(16))) float buf[256 + 1024];
    const int tid = threadIdx.x;   // one wave
    for (int i = tid; i < 256 + 1024; i += 64) buf[i] = 7.0f;   // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n_floats * 4, 0x00020000);
    // dword: lanes 0..31 in range (element = lane), lanes 32..47 just beyond the end, lanes 48..63 at 0x80000000
    int voff = tid < 32 ? tid * 4 : (tid < 48 ? (n_floats + tid - 32) * 4 : (int)0x80000000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)buf, 4, voff, 0, 0, 0);
    // dwordx4: lane l -> 16 bytes at element 4*l; lanes 32.. out of range; soffset moves everything by 64 bytes
    int voff4 = tid < 32 ? tid * 16 : (int)0x80000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(buf + 256), 16, voff4, 64, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = tid; i < 256 + 1024; i += 64) out[i] = buf[i];
}

int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 100.f + i;
    float *x, *out;
    hipMalloc(&x, 2 * n * sizeof(float));     // allocation larger than the declared buffer
    hipMalloc(&out, (256 + 1024) * sizeof(float));
    std::vector<float> h2(2 * n, 55.f);
    for (int i = 0; i < n; ++i) h2[i] = h[i];
    hipMemcpy(x, h2.data(), 2 * n * sizeof(float), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(x, n, out);
    std::vector<float> o(256 + 1024);
    hipMemcpy(o.data(), out, o.size() * sizeof(float), hipMemcpyDeviceToHost);
    printf("dword  in-range lanes 0..3 : %g %g %g %g (expect 100 101 102 103)\n", o[0], o[1], o[2], o[3]);
    printf("dword  beyond-end lanes 32..35: %g %g %g %g (0 = zero written, 7 = LDS untouched, 55 = no range check)\n", o[32], o[33], o[34], o[35]);
    printf("dword  0x80000000 lanes 48..51: %g %g %g %g\n", o[48], o[49], o[50], o[51]);
    printf("dwordx4 lane 0 (soffset 64 B = 16 floats): %g %g %g %g (expect 116 117 118 119)\n", o[256], o[257], o[258], o[259]);
    printf("dwordx4 lane 31: %g %g %g %g (expect 240..243)\n", o[256 + 124], o[256 + 125], o[256 + 126], o[256 + 127]);
    printf("dwordx4 out-of-range lane 32: %g %g %g %g\n", o[256 + 128], o[256 + 129], o[256 + 130], o[256 + 131]);
    printf("dwordx4 out-of-range lane 63: %g %g %g %g\n", o[256 + 252], o[256 + 253], o[256 + 254], o[256 + 255]);
    printf("untouched tail: %g %g\n", o[256 + 256], o[256 + 1023]);
    return 0;
}
