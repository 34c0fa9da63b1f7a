/* tools/lzma2_stats.c -- RESEARCH TOOL (not product): decodes an LZMA2 stream with oracle/lzma2_dec.c and prints, per symbol
 * kind (literal, match, short rep, rep0..3), how many symbols there are, how many bytes they cover and how many bits the range
 * coder spent on them (sum of -log2 p of the coded bits), plus histograms of match lengths.  Used to compare the reference's
 * parse with the GPU path's on the same input.
 *   gcc -O2 -o /tmp/lzstats tools/lzma2_stats.c -lm && /tmp/lzstats stream.lzma2 <prop> <rawSize>
 */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
static double g_bits_pending; static double g_bits[7]; static unsigned long g_n[7], g_bytes[7], g_lenh[7][8];
static float g_cost[2049];
#define GCO_ON_BIT(p, bit) (g_bits_pending += (bit) ? g_cost[2048 - (p)] : g_cost[(p)])
#define GCO_ON_DIRECT(n) (g_bits_pending += (n))
#define GCO_ON_SYM(kind, len, dist) do { g_bits[kind] += g_bits_pending; g_bits_pending = 0; g_n[kind]++; g_bytes[kind] += (len); \
    g_lenh[kind][(len) < 2 ? 0 : (len) == 2 ? 1 : (len) == 3 ? 2 : (len) < 6 ? 3 : (len) < 10 ? 4 : (len) < 18 ? 5 : (len) < 64 ? 6 : 7]++; } while (0)
#include "../oracle/lzma2_dec.c"
int main(int argc, char** argv)
{
    static const char* nm[7] = { "literal", "match", "shortrep", "rep0", "rep1", "rep2", "rep3" };
    for (int i = 1; i <= 2048; i++) g_cost[i] = (float)(-log2((double)i / 2048.0));
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* in = malloc(n); if (fread(in, 1, n, f) != n) return 1; fclose(f);
    size_t cap = strtoull(argv[3], 0, 10); uint8_t* out = malloc(cap + 16);
    size_t r = gco_lzma2_decode(in, n, out, cap, (unsigned char)atoi(argv[2]));
    printf("decoded %zu of %zu bytes from %zu\n", r, cap, n);
    double tb = 0; for (int k = 0; k < 7; k++) tb += g_bits[k];
    printf("%-9s %10s %11s %12s %7s %7s   len: 1 2 3 4-5 6-9 10-17 18-63 64+\n", "kind", "symbols", "bytes", "coded bytes", "share", "bit/sym");
    for (int k = 0; k < 7; k++) { printf("%-9s %10lu %11lu %12.0f %6.1f%% %7.2f  ", nm[k], g_n[k], g_bytes[k], g_bits[k] / 8, 100 * g_bits[k] / tb, g_n[k] ? g_bits[k] / g_n[k] : 0);
        for (int j = 0; j < 8; j++) printf(" %lu", g_lenh[k][j]); printf("\n"); }
    printf("total coded bytes %.0f\n", tb / 8);
    return 0;
}
