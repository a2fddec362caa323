// cycles per 32-cell block of the POA row kernel's body on L1-resident data (what does the instruction mix alone cost on this CPU?)
//   g++ -O3 -std=c++17 -mavx512f -mavx512bw tools/micro/poa_block.cpp -o /tmp/poa_block
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
alignas(64) static int16_t P[2][512 + 64], SC[512 + 64];
int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    for (int i = 0; i < 512 + 64; i++) { P[0][i] = (int16_t)(i * 7 % 100); P[1][i] = 0; SC[i] = (int16_t)((i * 13 % 4) ? -2 : 9); }
    const __m512i NEGV = _mm512_set1_epi16((short)-32768);
    alignas(64) static const short SHR1[32] = {0,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30};
    alignas(64) static const short PRV1[32] = {31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62};
    const __m512i floorv = _mm512_set1_epi16((short)-30000), gv = _mm512_set1_epi16((short)-6), i1 = _mm512_load_si512(SHR1), ip1 = _mm512_load_si512(PRV1), last = _mm512_set1_epi16(31), dv = _mm512_set1_epi16(9);
    const int rows = 2000000, nb = 10;
    if (variant == 7) {
        // STRIPED row (Farrar layout): V vectors per row, lane l of vector v holds column l*V + v.  The insertion chain (prefix maximum in the
        // ramped frame) is V-1 in-lane maxima + ONE 32-lane scan of the lane totals + V maxima, instead of a 5-step scan per block.
        const int V = 11;
        alignas(64) static int16_t R[2][11 * 32], PS[11 * 32];
        for (int i = 0; i < V * 32; i++) { R[0][i] = (int16_t)(i * 7 % 100); R[1][i] = 0; PS[i] = (int16_t)((i * 13 % 4) ? -2 : 9); }
        alignas(64) static const short SHL1[32] = {0,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30};
        const __m512i sh1 = _mm512_load_si512(SHL1);
        auto t0 = std::chrono::steady_clock::now();
        long sink = 0;
        for (int r = 0; r < rows; r++) {
            const int16_t* p = R[r & 1]; int16_t* row = R[(r & 1) ^ 1];
            __m512i m[11];
            // diag of vector 0 = predecessor's vector V-1 one lane down (column j-1 of lane l's first column is lane l-1's last column)
            __m512i pprev = _mm512_mask_permutexvar_epi16(NEGV, 0xFFFFFFFEu, sh1, _mm512_load_si512(p + (V - 1) * 32));
            __m512i run = NEGV;
            #pragma GCC unroll 11
            for (int v = 0; v < V; v++) {
                const __m512i pcur = _mm512_load_si512(p + v * 32);
                const __m512i d = _mm512_adds_epi16(pprev, _mm512_load_si512(PS + v * 32));
                const __m512i u = _mm512_adds_epi16(pcur, gv);
                pprev = pcur;
                const __m512i x = _mm512_max_epi16(floorv, _mm512_adds_epi16(_mm512_max_epi16(d, u), dv));
                run = _mm512_max_epi16(run, x);          // in-lane prefix maximum over the vectors
                m[v] = run;
            }
            // exclusive prefix maximum of the lane totals across the 32 lanes
            __m512i e = _mm512_mask_permutexvar_epi16(NEGV, 0xFFFFFFFEu, sh1, run);
            e = _mm512_max_epi16(e, _mm512_mask_permutexvar_epi16(NEGV, 0xFFFFFFFEu, i1, e));
            e = _mm512_max_epi16(e, _mm512_alignr_epi32(e, NEGV, 15));
            e = _mm512_max_epi16(e, _mm512_alignr_epi32(e, NEGV, 14));
            e = _mm512_max_epi16(e, _mm512_alignr_epi32(e, NEGV, 12));
            e = _mm512_max_epi16(e, _mm512_alignr_epi32(e, NEGV, 8));
            #pragma GCC unroll 11
            for (int v = 0; v < V; v++) _mm512_store_si512(row + v * 32, _mm512_max_epi16(m[v], e));
            sink += row[5] + row[100] + row[319];
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("variant 7 (striped, V = %d = 352 slots): %.2f ns per row (sink %ld)\n", V, dt / rows * 1e9, sink);
        return 0;
    }
    if (variant == 5 || variant == 6) {
        const int NR = 2000, RS = 336;                       // matrix rows per alignment, row stride (elements): 1.34 MB of int16
        static int16_t* M = (int16_t*)aligned_alloc(64, (size_t)NR * RS * 2 + 4096);
        static uint8_t* B = (uint8_t*)aligned_alloc(64, (size_t)NR * RS + 4096);
        for (size_t i = 0; i < (size_t)NR * RS; i++) { M[i] = (int16_t)(i * 7 % 100); B[i] = 0; }
        auto t0 = std::chrono::steady_clock::now();
        long sink = 0;
        const int aligns = rows / NR;
        for (int a = 0; a < aligns; a++) {
            for (int r = 1; r < NR; r++) {
                const int16_t* p; int16_t* row;
                if (variant == 5) { p = M + (size_t)(r - 1) * RS + 8; row = M + (size_t)r * RS + 8 + 1; }       // the band shifts by one column per row, as in the real matrix
                else { p = M + (size_t)((r - 1) & 15) * RS + 8; row = M + (size_t)(r & 15) * RS + 8 + 1; }
                __m512i carry = NEGV;
                __m512i pprev = _mm512_loadu_si512(p - 32 + 32);
                for (int b = 0; b < nb; b++) {
                    const int j = 32 * b;
                    const __m512i pcur = _mm512_loadu_si512(p + j + 1);
                    const __m512i d = _mm512_adds_epi16(_mm512_loadu_si512(p + j), _mm512_loadu_si512(SC + j));
                    const __m512i u = _mm512_adds_epi16(pcur, gv);
                    pprev = pcur;
                    __m512i x = _mm512_max_epi16(floorv, _mm512_adds_epi16(_mm512_max_epi16(d, u), dv));
                    const __m512i x0 = x;
                    x = _mm512_max_epi16(x, _mm512_mask_permutexvar_epi16(NEGV, 0xFFFFFFFEu, i1, x));
                    x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 15));
                    x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 14));
                    x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 12));
                    x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 8));
                    const __m512i lastv = _mm512_permutexvar_epi16(last, x);
                    const __m512i out = _mm512_max_epi16(x, carry);
                    _mm512_storeu_si512(row + j, out);
                    carry = _mm512_max_epi16(carry, lastv);
                    if (variant == 6) {                      // back-pointer bytes: diag / up / left by comparing the candidates with the result
                        const __mmask32 md = _mm512_cmpeq_epi16_mask(_mm512_adds_epi16(d, dv), out), mu = _mm512_cmpeq_epi16_mask(_mm512_adds_epi16(u, dv), out);
                        __m256i bp = _mm256_maskz_set1_epi8(mu, 2);
                        bp = _mm256_mask_set1_epi8(bp, md, 1);
                        _mm256_storeu_si256((__m256i*)(B + (size_t)r * RS + j), bp);
                        (void)x0;
                    }
                }
                sink += row[5];
            }
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("variant %d: %.2f ns per row (%d alignments of %d rows; %s)\n", variant, dt / ((double)aligns * (NR - 1)) * 1e9, aligns, NR, variant == 5 ? "rows stream through a 1.3 MB matrix" : "rows in a 16-row ring + a byte per cell streamed");
        return (int)(sink & 1);
    }
    auto t0 = std::chrono::steady_clock::now();
    long sink = 0;
    for (int r = 0; r < rows; r++) {
        const int16_t* p = P[r & 1] + 32; int16_t* row = P[(r & 1) ^ 1] + 32;
        __m512i carry = NEGV;
        __m512i pprev = _mm512_loadu_si512(p - 32);
        for (int b = 0; b < nb; b++) {
            const int j = 32 * b;
            const __m512i pcur = _mm512_load_si512(p + j);
            __m512i x;
            if (variant == 3) x = _mm512_adds_epi16(pcur, _mm512_load_si512(SC + j));
            else {
                const __m512i d = _mm512_adds_epi16(_mm512_permutex2var_epi16(pprev, ip1, pcur), _mm512_load_si512(SC + j));
                const __m512i u = _mm512_adds_epi16(pcur, gv);
                pprev = pcur;
                x = _mm512_max_epi16(floorv, _mm512_adds_epi16(_mm512_max_epi16(d, u), dv));
            }
            if (variant == 4) {
                // in-lane prefix maximum on biased (unsigned) values: zero fill of the byte shifts is neutral; then the three lanes' maxima cross lanes
                const __m512i BIAS = _mm512_set1_epi16((short)0x8000);
                __m512i y = _mm512_xor_si512(x, BIAS);
                y = _mm512_max_epu16(y, _mm512_bslli_epi128(y, 2));
                y = _mm512_max_epu16(y, _mm512_bslli_epi128(y, 4));
                y = _mm512_max_epu16(y, _mm512_bslli_epi128(y, 8));
                const __m512i ZERO = _mm512_setzero_si512();
                __m512i t = _mm512_shuffle_epi8(y, _mm512_set4_epi32(0x0F0E0F0E, 0x0F0E0F0E, 0x0F0E0F0E, 0x0F0E0F0E));   // every lane's last element, broadcast in the lane
                t = _mm512_alignr_epi32(t, ZERO, 12);                                   // one 128-bit lane to the right: [0, t0, t1, t2]
                t = _mm512_max_epu16(t, _mm512_alignr_epi32(t, ZERO, 12));              // [0, t0, max(t0,t1), max(t1,t2)]
                t = _mm512_max_epu16(t, _mm512_alignr_epi32(t, ZERO, 8));               // [0, t0, max(t0,t1), max(t0,t1,t2)]
                y = _mm512_max_epu16(y, t);
                x = _mm512_xor_si512(y, BIAS);
            } else
            if (variant != 1 && variant != 3) {
                x = _mm512_max_epi16(x, _mm512_mask_permutexvar_epi16(NEGV, 0xFFFFFFFEu, i1, x));
                x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 15));
                x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 14));
                x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 12));
                x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 8));
            }
            const __m512i lastv = _mm512_permutexvar_epi16(last, x);
            _mm512_store_si512(row + j, _mm512_max_epi16(x, carry));
            carry = _mm512_max_epi16(carry, lastv);
        }
        if (variant == 2) { P[(r & 1) ^ 1][40] = 0; }
        sink += row[5] + row[100] + row[319];
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("variant %d: %.2f ns per row of %d blocks, %.2f ns per block (sink %ld)\n", variant, dt / rows * 1e9, nb, dt / rows / nb * 1e9, sink);
    return 0;
}
