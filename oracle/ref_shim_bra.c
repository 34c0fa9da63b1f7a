/* oracle/ref_shim_bra.c -- TEST INFRASTRUCTURE ONLY.
 * The REFERENCE's branch converters (compiled from /root/reference/C/Bra.c by oracle/Makefile into oracle/_ref/libbra_ref.so), one entry point
 * that converts a buffer in place with one call, the way NCompress::NBranch::CCoder::Filter does (CPP/7zip/Compress/BranchMisc.cpp:21-26). */
#include <stddef.h>
#include "Bra.h"
#include "Delta.h"
#include "7zCrc.h"

/* kind: 0 ARM64, 1 ARM, 2 ARMT, 3 PPC, 4 SPARC, 5 IA64, 6 RISCV; returns the processed byte count */
size_t ref_bra_convert(int kind, unsigned char* data, size_t n, unsigned pc, int encoding)
{
    z7_Func_BranchConv f = 0;
    switch (kind) {
        case 0: f = encoding ? z7_BranchConv_ARM64_Enc : z7_BranchConv_ARM64_Dec; break;
        case 1: f = encoding ? z7_BranchConv_ARM_Enc : z7_BranchConv_ARM_Dec; break;
        case 2: f = encoding ? z7_BranchConv_ARMT_Enc : z7_BranchConv_ARMT_Dec; break;
        case 3: f = encoding ? z7_BranchConv_PPC_Enc : z7_BranchConv_PPC_Dec; break;
        case 4: f = encoding ? z7_BranchConv_SPARC_Enc : z7_BranchConv_SPARC_Dec; break;
        case 5: f = encoding ? z7_BranchConv_IA64_Enc : z7_BranchConv_IA64_Dec; break;
        case 6: f = encoding ? z7_BranchConv_RISCV_Enc : z7_BranchConv_RISCV_Dec; break;
        default: return (size_t)-1;
    }
    return (size_t)(f(data, n, pc) - data);
}

/* X86 with its state (C/Bra86.c): returns the processed byte count, *state in/out */
size_t ref_bra_x86_convert(unsigned char* data, size_t n, unsigned pc, int encoding, unsigned* state)
{
    UInt32 st = *state;
    Byte* e = encoding ? z7_BranchConvSt_X86_Enc(data, n, pc, &st) : z7_BranchConvSt_X86_Dec(data, n, pc, &st);
    *state = st;
    return (size_t)(e - data);
}

/* Delta filter (C/Delta.c) in place; state[256] in/out */
void ref_delta_convert(unsigned char* data, size_t n, unsigned delta, int encoding, unsigned char* state)
{
    if (encoding) Delta_Encode(state, delta, data, n); else Delta_Decode(state, delta, data, n);
}

/* CRC-32 as the 7z container computes it (C/7zCrc.c: CrcCalc after CrcGenerateTable) */
unsigned ref_crc32(const unsigned char* data, size_t n)
{
    static int ready = 0;
    if (!ready) { CrcGenerateTable(); ready = 1; }
    return CrcCalc(data, n);
}
