// tools/hazard_repro.hip -- minimal reproducer of the gfx950 store-data hazard.
// Every lane stores 16 bytes with `buffer_store_dwordx4 v[10:13], voff, rsrc, soff offen` and the very
// next instruction(s) overwrite v12 / v13 with a poison value.  Architecturally the store must write the
// OLD values.  Variants: soffset in an SGPR or the literal 0; 0, 1 or 2 wait states (s_nop) in between.
// The host counts poisoned dwords in memory.  hipcc -O3 --offload-arch=gfx950 -o hazard_repro hazard_repro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int SGPR_SOFF, int NOPS>
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned n_per_block, int rounds) {
    // one buffer descriptor per block: base = out + blockIdx.x * n_per_block dwords
    const unsigned long long base = (unsigned long long)(out + (size_t)blockIdx.x * n_per_block);
    i32x4 rsrc;
    rsrc.x = (int)(unsigned)base;
    rsrc.y = (int)(unsigned)(base >> 32);
    rsrc.z = (int)(n_per_block * 4u);
    rsrc.w = 0x00020000;
    const unsigned poison = 0xDEADBEEFu;
    for (int r = 0; r < rounds; ++r) {
        const unsigned voff = (threadIdx.x + 256u * (unsigned)r) * 16u;
        const unsigned tag = blockIdx.x * 65536u + r * 256u + threadIdx.x;
        const unsigned soff = 0u;   // value 0 either way: only the encoding (SGPR vs literal) differs
        if (SGPR_SOFF) {
            asm volatile(
                "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n"
                "s_nop 4\n"
                "buffer_store_dwordx4 v[10:13], %1, %2, %3 offen\n"
                ".if %c5 == 1\n s_nop 0\n .endif\n .if %c5 == 2\n s_nop 1\n .endif\n"
                "v_mov_b32 v12, %4\n v_mov_b32 v13, %4\n v_mov_b32 v10, %4\n v_mov_b32 v11, %4\n"
                : : "v"(tag), "v"(voff), "s"(rsrc), "s"(soff), "v"(poison), "n"(NOPS) : "v10", "v11", "v12", "v13", "memory");
        } else {
            asm volatile(
                "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n"
                "s_nop 4\n"
                "buffer_store_dwordx4 v[10:13], %1, %2, 0 offen\n"
                ".if %c4 == 1\n s_nop 0\n .endif\n .if %c4 == 2\n s_nop 1\n .endif\n"
                "v_mov_b32 v12, %3\n v_mov_b32 v13, %3\n v_mov_b32 v10, %3\n v_mov_b32 v11, %3\n"
                : : "v"(tag), "v"(voff), "s"(rsrc), "v"(poison), "n"(NOPS) : "v10", "v11", "v12", "v13", "memory");
        }
    }
}

// the same with a global store (64-bit VGPR address), the form hipcc uses for plain float4 stores
template <int NOPS>
__global__ void __launch_bounds__(256) kg(unsigned* out, unsigned n_per_block, int rounds) {
    unsigned* base = out + (size_t)blockIdx.x * n_per_block;
    const unsigned poison = 0xDEADBEEFu;
    for (int r = 0; r < rounds; ++r) {
        unsigned* p = base + (threadIdx.x + 256u * (unsigned)r) * 4u;
        const unsigned tag = blockIdx.x * 65536u + r * 256u + threadIdx.x;
        asm volatile(
            "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n"
            "s_nop 4\n"
            "global_store_dwordx4 %1, v[10:13], off\n"
            ".if %c3 == 1\n s_nop 0\n .endif\n .if %c3 == 2\n s_nop 1\n .endif\n"
            "v_mov_b32 v12, %2\n v_mov_b32 v13, %2\n v_mov_b32 v10, %2\n v_mov_b32 v11, %2\n"
            : : "v"(tag), "v"(p), "v"(poison), "n"(NOPS) : "v10", "v11", "v12", "v13", "memory");
    }
}
// 8-byte buffer store with an SGPR soffset, overwritten at once (the library's half-precision / 512-thread paths)
__global__ void __launch_bounds__(256) k2(unsigned* out, unsigned n_per_block, int rounds) {
    const unsigned long long base = (unsigned long long)(out + (size_t)blockIdx.x * n_per_block);
    i32x4 rsrc;
    rsrc.x = (int)(unsigned)base; rsrc.y = (int)(unsigned)(base >> 32); rsrc.z = (int)(n_per_block * 4u); rsrc.w = 0x00020000;
    const unsigned poison = 0xDEADBEEFu, soff = 0u;
    for (int r = 0; r < rounds; ++r) {
        const unsigned voff = (threadIdx.x + 256u * (unsigned)r) * 16u;
        const unsigned tag = blockIdx.x * 65536u + r * 256u + threadIdx.x;
        asm volatile(
            "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n s_nop 4\n"
            "buffer_store_dwordx2 v[10:11], %1, %2, %3 offen\n"
            "v_mov_b32 v11, %4\n v_mov_b32 v10, %4\n"
            : : "v"(tag), "v"(voff), "s"(rsrc), "s"(soff), "v"(poison) : "v10", "v11", "memory");
    }
}
long run2(unsigned* d, std::vector<unsigned>& h, int blocks, int rounds, int reps) {
    const unsigned n_per_block = 256u * 4u * (unsigned)rounds;
    long bad = 0;
    for (int i = 0; i < reps; ++i) {
        (void)hipMemset(d, 0, h.size() * 4);
        k2<<<blocks, 256>>>(d, n_per_block, rounds);
        (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        for (size_t j = 0; j < h.size(); ++j) bad += h[j] == 0xDEADBEEFu;
    }
    return bad;
}

template <int N> long rung(unsigned* d, std::vector<unsigned>& h, int blocks, int rounds, int reps) {
    const unsigned n_per_block = 256u * 4u * (unsigned)rounds;
    long bad = 0;
    for (int i = 0; i < reps; ++i) {
        (void)hipMemset(d, 0, h.size() * 4);
        kg<N><<<blocks, 256>>>(d, n_per_block, rounds);
        (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        for (size_t j = 0; j < h.size(); ++j) bad += h[j] == 0xDEADBEEFu;
    }
    return bad;
}

template <int S, int N> long run(unsigned* d, std::vector<unsigned>& h, int blocks, int rounds, int reps) {
    const unsigned n_per_block = 256u * 4u * (unsigned)rounds;
    long bad = 0;
    for (int i = 0; i < reps; ++i) {
        (void)hipMemset(d, 0, h.size() * 4);
        k<S, N><<<blocks, 256>>>(d, n_per_block, rounds);
        (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        for (size_t j = 0; j < h.size(); ++j) bad += h[j] == 0xDEADBEEFu;
    }
    return bad;
}
int main() {
    const int blocks = 256 * 16, rounds = 16, reps = 20;
    std::vector<unsigned> h((size_t)blocks * 256 * 4 * rounds);
    unsigned* d; if (hipMalloc(&d, h.size() * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const double total = (double)h.size() * reps;
    printf("dwords stored per variant: %.3g\n", total);
    printf("SGPR soffset, 0 wait states: %ld poisoned dwords\n", run<1, 0>(d, h, blocks, rounds, reps));
    printf("SGPR soffset, 1 wait state : %ld\n", run<1, 1>(d, h, blocks, rounds, reps));
    printf("SGPR soffset, 2 wait states: %ld\n", run<1, 2>(d, h, blocks, rounds, reps));
    printf("literal soffset 0, 0 wait states: %ld\n", run<0, 0>(d, h, blocks, rounds, reps));
    printf("literal soffset 0, 1 wait state : %ld\n", run<0, 1>(d, h, blocks, rounds, reps));
    printf("literal soffset 0, 2 wait states: %ld\n", run<0, 2>(d, h, blocks, rounds, reps));
    printf("buffer_store_dwordx2, SGPR soffset, 0 wait states: %ld\n", run2(d, h, blocks, rounds, reps));
    printf("global_store_dwordx4, 0 wait states: %ld\n", rung<0>(d, h, blocks, rounds, reps));
    printf("global_store_dwordx4, 1 wait state : %ld\n", rung<1>(d, h, blocks, rounds, reps));
    printf("global_store_dwordx4, 2 wait states: %ld\n", rung<2>(d, h, blocks, rounds, reps));
    return 0;
}
