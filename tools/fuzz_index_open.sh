#!/bin/bash
# tools/fuzz_index_open.sh [SEED] [ITERATIONS] -- CPU only: builds csrc/mrk_files.cpp alone with ASan + UBSan and feeds
# mrk_index_open (and mrk_rt_ram_open: RT RAM chunks) mutated copies of the index fixtures under tests/golden/indexes/ (bytes flipped, files cut, junk inserted,
# and -- for the header's count fields: n_fields, n_attrs, n_checkpoints, m_iDocinfo, embedded-list counts -- dwords / qwords
# at random header offsets overwritten with extreme values such as 0x40000000 or 2^62 + 1).
# Every open must end in MRK_OK or an error code; the sanitizers abort on anything else.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/mrk_fuzz_open
mkdir -p $W
cat > $W/stub.cpp <<'CPP'
#include <stdarg.h>
#include <stdio.h>
#include "mrk_hostindex.h"
static char g_err[1024];
int mrk_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return code; }
extern "C" void fuzz_free(mrk_host_index* h) { delete h; }
CPP
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -I$ROOT/manticoresearch_amd/csrc \
    $ROOT/manticoresearch_amd/csrc/mrk_files.cpp $ROOT/manticoresearch_amd/csrc/mrk_writer.cpp $W/stub.cpp -lpthread -o $W/libfiles_asan.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python3 - "$ROOT" "$W" "${1:-1}" "${2:-4000}" <<'PY'
import ctypes as C, os, random, sys
root, w, seed, n_iter = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
L = C.CDLL(w + "/libfiles_asan.so")
L.mrk_index_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
L.mrk_host_index_find_word.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
L.mrk_host_index_word.restype = C.c_void_p
L.mrk_host_index_word.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
L.fuzz_free.argtypes = [C.c_void_p]
random.seed(seed)
src = root + "/tests/golden/indexes/"
names = ["t233_reload", "t250_plain2", "t406_index0", "t233_test"]
ok = err = 0
for _ in range(n_iter):
    name = random.choice(names)
    for e in ("sph", "spi", "spd", "spp", "spe", "spm", "spa"):
        data = bytearray(open(src + name + "." + e, "rb").read())
        if random.random() < 0.6:
            for _ in range(random.randint(1, 5)):
                if not data:
                    break
                k, r = random.randrange(len(data)), random.random()
                if r < 0.5:
                    data[k] = random.choice([0, 1, 0x7F, 0x80, 0xFF, random.randrange(256)])
                elif r < 0.75:
                    del data[k:]
                else:
                    data[k:k] = bytes(random.randrange(256) for _ in range(random.randint(1, 8)))
        if e == "sph" and data and random.random() < 0.5:  # extreme counts: every dword / qword of the header gets its turn
            import struct
            for _ in range(random.randint(1, 2)):
                k = random.randrange(len(data))
                if random.random() < 0.6:
                    v = random.choice([0xFFFFFFFF, 0x40000000, 0x08000000, 0x7FFFFFFF, 0x80000000, 0x10000, 65537, 257])
                    data[k:k + 4] = struct.pack("<I", v)
                else:
                    v = random.choice([(1 << 62) + 1, (1 << 63), (1 << 64) - 1, 1 << 32, (1 << 40) + 7])
                    data[k:k + 8] = struct.pack("<Q", v)
        open(w + "/x." + e, "wb").write(bytes(data))
    h = C.c_void_p()
    if L.mrk_index_open((w + "/x").encode(), C.byref(h)) == 0:
        ok += 1
        L.mrk_host_index_find_word(h, b"index", 5)
        n = C.c_uint32()
        for t in range(8):
            L.mrk_host_index_word(h, t, C.byref(n))
        L.fuzz_free(h)
    else:
        err += 1
print("opened", ok, "rejected", err, "-- no sanitizer report")
# RT RAM chunks (mrk_rt_ram_open): the same treatment for <prefix>.meta + <prefix>.ram
L.mrk_rt_ram_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
L.mrk_rt_ram_segments.argtypes = [C.c_void_p]
L.mrk_rt_ram_free.argtypes = [C.c_void_p]
import struct
ok = err = 0
for _ in range(n_iter):
    name = random.choice(["t406_idx320", "t406_index"])
    for e in ("meta", "ram"):
        data = bytearray(open(src + name + "." + e, "rb").read())
        if random.random() < 0.7:
            for _ in range(random.randint(1, 4)):
                if not data:
                    break
                k, r = random.randrange(len(data)), random.random()
                if r < 0.4:
                    data[k] = random.choice([0, 1, 0x7F, 0x80, 0xFF, random.randrange(256)])
                elif r < 0.55:
                    del data[k:]
                elif r < 0.7:
                    data[k:k] = bytes(random.randrange(256) for _ in range(random.randint(1, 8)))
                else:
                    data[k:k + 4] = struct.pack("<I", random.choice([0xFFFFFFFF, 0x40000000, 0x7FFFFFFF, 0x80000000, 0x10000, 257]))
        open(w + "/y." + e, "wb").write(bytes(data))
    rt = C.c_void_p()
    if L.mrk_rt_ram_open((w + "/y").encode(), C.byref(rt)) == 0:
        ok += 1
        L.mrk_rt_ram_segments(rt)
        L.mrk_rt_ram_free(rt)
    else:
        err += 1
# fixed cases (round 3): a segment's row count chosen so that (rows + 31) / 32 wraps in 32 bits, with the dead-row map in place and with
# every 4-byte piece behind the header removed in turn
fixed = 0
for name in ("t406_idx320", "t406_index"):
    meta, ram = open(src + name + ".meta", "rb").read(), open(src + name + ".ram", "rb").read()
    open(w + "/z.meta", "wb").write(meta)
    for rows in (0xFFFFFFF0, 0xFFFFFFFF, 0xFFFFFFE1, 0x80000000, 0x7FFFFFFF):
        for cut in [None] + list(range(12, len(ram) - 4, 4)):
            rb = bytearray(ram)
            rb[8:12] = struct.pack("<I", rows)
            if cut is not None:
                del rb[cut:cut + 4]
            open(w + "/z.ram", "wb").write(bytes(rb))
            rt = C.c_void_p()
            if L.mrk_rt_ram_open((w + "/z").encode(), C.byref(rt)) == 0:
                L.mrk_rt_ram_free(rt)
            fixed += 1
print("RT RAM chunks: opened", ok, "rejected", err, "+", fixed, "fixed row-count cases -- no sanitizer report")
PY
