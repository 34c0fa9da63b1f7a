#!/bin/bash
# oracle/build_ref_7z.sh -- TEST INFRASTRUCTURE: build the REFERENCE's own console host (`7z`, built with Z7_EXTERNAL_CODECS)
# and its format/codec bundle (`7z.so`) from the sources where they lie under /root/reference, into oracle/_ref/host7z/ (git-ignored,
# travels with gpurun like the other reference binaries).  The reference tree is only read: its makefiles write their objects next
# to the sources, so they run on a scratch copy under /tmp that is deleted afterwards; only the two binaries are kept.  Used by tests/test_real_host.py to drive the plugin the way the product's
# host does (config C1: `7z a -m0=<method> -mx<level>` + `7z t`; CPP/7zip/UI/Common/LoadCodecs.cpp:531-650 loads Codecs/*.so).
# A second bundle, 7z_nozstd.so, is the same link without ZstdRegister.o: a host that has no ZSTD codec of its own (like mainline 7-Zip), so that
# `7z x` / `7z t` of a ZSTD archive resolve the method id to the plugin's DECODER (CreateCoder.cpp:206-232 asks built-in codecs first).
# A third one, 7z_nobrotli.so, is the link without BrotliRegister.o (the brotli library itself stays inside: the plugin takes the static dictionary of RFC 7932 from the host's
# BrotliGetDictionary): `7z x` / `7z t` of a BROTLI archive resolve the method to the plugin's decoder.
# usage: build_ref_7z.sh <outdir>      -> <outdir>/7z, <outdir>/7z.so, <outdir>/7z_nozstd.so, <outdir>/7z_nobrotli.so
set -e
REF=${REF_ROOT:-/root/reference}
OUT=${1:?outdir}
[ -x "$OUT/7z" ] && [ -f "$OUT/7z.so" ] && [ -f "$OUT/7z_nozstd.so" ] && [ -f "$OUT/7z_nobrotli.so" ] && exit 0
[ -d "$REF/CPP/7zip/UI/Console" ] || { echo "no reference tree at $REF" >&2; exit 3; }
mkdir -p "$OUT"
SCR=$(mktemp -d /tmp/gc_ref7z.XXXXXX)
trap 'rm -rf "$SCR"' EXIT
cp -r "$REF/C" "$REF/CPP" "$SCR/"; [ -d "$REF/Asm" ] && cp -r "$REF/Asm" "$SCR/"
J=${JOBS:-$(nproc)}
( cd "$SCR/CPP/7zip/UI/Console" && make -j$J -f makefile.gcc > "$OUT/build_console.log" 2>&1 )
( cd "$SCR/CPP/7zip/Bundles/Format7zF" && make -j$J -f makefile.gcc > "$OUT/build_format7zf.log" 2>&1 )
cp "$SCR/CPP/7zip/UI/Console/_o/7z" "$OUT/7z"
cp "$SCR/CPP/7zip/Bundles/Format7zF/_o/7z.so" "$OUT/7z.so"
sed -i '/ZstdRegister\.o/d' "$SCR/CPP/7zip/Bundles/Format7zF/Arc_gcc.mak"
rm -f "$SCR/CPP/7zip/Bundles/Format7zF/_o/7z.so"
( cd "$SCR/CPP/7zip/Bundles/Format7zF" && make -j$J -f makefile.gcc > "$OUT/build_format7zf_nozstd.log" 2>&1 )
cp "$SCR/CPP/7zip/Bundles/Format7zF/_o/7z.so" "$OUT/7z_nozstd.so"
cp "$REF/CPP/7zip/Bundles/Format7zF/Arc_gcc.mak" "$SCR/CPP/7zip/Bundles/Format7zF/Arc_gcc.mak"
sed -i '/BrotliRegister\.o/d' "$SCR/CPP/7zip/Bundles/Format7zF/Arc_gcc.mak"
rm -f "$SCR/CPP/7zip/Bundles/Format7zF/_o/7z.so"
( cd "$SCR/CPP/7zip/Bundles/Format7zF" && make -j$J -f makefile.gcc > "$OUT/build_format7zf_nobrotli.log" 2>&1 )
cp "$SCR/CPP/7zip/Bundles/Format7zF/_o/7z.so" "$OUT/7z_nobrotli.so"
