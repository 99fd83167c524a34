// attic: the L2 warm-up of the next two mat-vecs by the spare workgroups of the decode attention launch (option "prefetch",
// round 2: 723 -> 730-733 tok/s at best, slower with more).  Withdrawn in round 4 with its four options:
// the attention launch it rode on is now part of k_qkv_attn.  Not compiled.
// ---------------------------------------------------------------------------------------------------
// L2 warm-up for the next two mat-vecs, done by the spare workgroups of the decode k_attn_decode launch (which keeps only
// n_head CUs busy on a latency chain and leaves HBM idle for its whole duration).  Two facts make it worth doing
// (tests/tools/overlap_probe3.hip, profiles/r02_overlap_probe3.txt): an XCD's L2 keeps its lines across a kernel boundary
// (a link whose first ring steps were requested by its predecessor sees its first weights 0.86 us after entry instead
// of 2.19), and workgroup b always runs on XCD b mod 8.  k_mmvq_big deals unit u to workgroup u mod G (G a multiple
// of 8), so the workgroups of XCD x read exactly the rows r = x (mod 8) — and the spare workgroup b pulls those rows
// (ordinary loads, results discarded) into the L2 the consumer will look them up in.  The first version of this option
// cut the planes into contiguous slices regardless of the XCD: the lines then sat in the memory-side Infinity Cache
// and in the wrong L2s, and the mat-vecs gained 0.4-0.6 us for 1.4 us more attention.
// ---------------------------------------------------------------------------------------------------
struct PrefetchArgs {
    const u32x4 *p[6];   // planes: 16-byte quants and f16 scales of wo, w1, w3
    int rows[6];         // leading rows of the plane to pull (0 = plane unused)
    int row16[6];        // 16-byte pieces per row
    int unit_rows[6];    // rows per dealt unit (1; 2 would be wq|wk|wv's row pairs)
    int delay;           // ~0.2 us units before the first request (the heads' own K/V requests go out first)
    unsigned *sink;      // never written (the condition below is never true), keeps the loads alive
};
// Workgroup `wg` (its blockIdx) of `G8` = gridDim.x / 8 workgroups per XCD class: 8 independent 16-byte loads per thread
// in flight, default cache policy.
__device__ __forceinline__ void prefetch_slice(const PrefetchArgs &a, int wg, int first_wg, int n_wg) {
    u32x4 acc = {0, 0, 0, 0};
    const int x = wg & 7;                                  // this workgroup's XCD = the consumer's
    const int rank = (wg - first_wg) >> 3, per = n_wg >> 3;  // position among the spare workgroups of that XCD
    if (rank >= per) return;                               // n_wg not a multiple of 8: the stragglers do nothing
    for (int i = 0; i < a.delay; i++) __builtin_amdgcn_s_sleep(8);  // 8 x 64 clocks ~ 0.2 us per unit
    const int CT = per * (int)blockDim.x, ct = rank * (int)blockDim.x + (int)threadIdx.x;
#pragma unroll 1
    for (int s = 0; s < 6; s++) {
        const int r16 = a.row16[s], ur = a.unit_rows[s];
        if (a.rows[s] <= 0) continue;
        const int units = a.rows[s] / ur;
        const int n_cls = units > x ? (units - x + 7) >> 3 : 0;  // units u = x (mod 8)
        const int n = n_cls * ur * r16;                          // 16-byte pieces of this class
        for (int q0 = ct; q0 < n; q0 += CT * 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int q = q0 + u * CT;
                const int j = q / (ur * r16), in = q - j * (ur * r16);  // class-local unit, piece within the unit
                const size_t off = (size_t)(x + 8 * j) * (size_t)(ur * r16) + (size_t)in;
                v[u] = q < n ? a.p[s][off] : u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc ^= v[u];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u && a.sink) a.sink[0] = acc[0];
}

