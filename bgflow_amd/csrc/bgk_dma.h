/* bgk_dma.h -- global -> LDS copies on the DMA path (global_load_lds: no staging registers, every request in flight from the first
 * cycle) for the kernels whose wave owns a contiguous [64][w] tile of a row-major tensor: the sampling tail / inference head
 * (bgk_tail.hip) and the IC -> xyz backward sweep (bgk_ic.hip).  Device only; include inside the translation unit's namespace. */
#ifndef BGK_DMA_H
#define BGK_DMA_H

typedef const __attribute__((address_space(1))) void* gvp_t;
typedef __attribute__((address_space(3))) void* lvp_t;

/* s_waitcnt vmcnt(c) for a run-time (wave-uniform) count: the instruction takes an immediate */
__device__ __forceinline__ void wait_vmcnt(int c) {
#define BGK_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (c) {
        BGK_W(3) BGK_W(6) BGK_W(9) BGK_W(12) BGK_W(15) BGK_W(18) BGK_W(21) BGK_W(24) BGK_W(27) BGK_W(30)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef BGK_W
}

/* linear DMA copy of the wave's [64][w] tile (256 w bytes, 16-byte aligned) into LDS; rows beyond `rows` re-read the tile's start */
__device__ __forceinline__ void dma_tile(float* dst, const float* __restrict__ src, int w, int rows, int lane) {
    const int valid = rows * w;                       /* floats */
    int c = 0;
    for (; c + 4 <= w; c += 4) {                      /* 1 KiB per instruction */
        const int e = c * 64 + lane * 4;
        __builtin_amdgcn_global_load_lds((gvp_t)(src + (e + 3 < valid ? e : 0)), (lvp_t)(dst + c * 64), 16, 0, 0);
    }
    const int covered = c * 64;                       /* floats the 16-byte requests above cover */
    for (; c < w; ++c) {                              /* 256 B per instruction */
        const int e = c * 64 + lane;
        __builtin_amdgcn_global_load_lds((gvp_t)(src + (e < valid ? e : 0)), (lvp_t)(dst + c * 64), 4, 0, 0);
    }
    /* a partial tile whose valid length is not a multiple of 4: the 16-byte piece straddling its end was redirected to the tile's
     * start as a whole (reading it would run past the tensor); its 1..3 valid floats follow here, one dword each (requests of a wave
     * land in order, so these overwrite the redirected piece) */
    const int edge = valid & ~3;
    if (edge < valid && edge < covered) {
        const int e = edge + lane;
        if (e < valid) __builtin_amdgcn_global_load_lds((gvp_t)(src + e), (lvp_t)(dst + edge), 4, 0, 0);
    }
}

#endif /* BGK_DMA_H */
