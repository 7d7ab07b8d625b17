#!/bin/bash
# integration/check_adapter.sh -- syntax-check integration/mrk_adapter.h against the reference's own headers.
# Build container only (needs /root/reference/src; nothing of it is copied or linked, no object is produced: -fsyntax-only).
# The reference's headers want the config.h its cmake would generate; for a syntax check the list of HAVE_* switches the
# survey found sufficient is written to a scratch directory.
set -e
REF=${REF:-/root/reference/src}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$REF" ] || { echo "check_adapter: $REF not present (build container only)"; exit 0; }
W=$(mktemp -d)
trap 'rm -rf $W' EXIT
for d in HAVE_CLOCK_GETTIME HAVE_DLOPEN HAVE_DLERROR HAVE_EPOLL HAVE_EVENTFD HAVE_MALLOC_TRIM HAVE_MALLOC_STATS HAVE_SO_REUSEPORT \
         HAVE_RWLOCK_PREFER_WRITER HAVE_PTHREAD_H HAVE_UNISTD_H HAVE_INTTYPES_H HAVE_STDINT_H HAVE_SYS_TYPES_H HAVE_SYNC_FETCH \
         HAVE_NANOSLEEP HAVE_PREAD HAVE_POLL HAVE_STRNLEN USE_LITTLE_ENDIAN UNALIGNED_RAM_ACCESS HAVE_PTHREAD_MUTEX_TIMEDLOCK \
         HAVE_PTHREAD_COND_TIMEDWAIT; do echo "#define $d 1"; done > $W/config.h
cat > $W/tu.cpp <<'CPP'
#include "mrk_adapter.h"
// instantiate what a caller would: the factory and the six ISphRanker methods
ISphRanker * (*g_fnCreate)( const XQQuery_t &, const CSphQuery &, CSphQueryResultMeta &, const ISphQwordSetup &, const CSphQueryContext &,
	const ISphSchema &, const VecTraits_T<ISphMatchSorter *> &, DWORD, const MrkIndexBinding_t &, mrk_batch *, mrk_batcher *, int, CSphString & ) = &MrkCreateRanker;
static_assert ( std::is_base_of<ISphRanker, MrkRankerAdapter_c>::value, "MrkRankerAdapter_c is an ISphRanker" );
static_assert ( !std::is_abstract<MrkRankerAdapter_c>::value, "every pure virtual of ISphRanker is implemented" );
CPP
g++ -std=c++14 -fsyntax-only -w -fpermissive -DHAVE_CONFIG_H -I$W -I$REF -I$ROOT/include -I$ROOT/integration $W/tu.cpp
echo "check_adapter: integration/mrk_adapter.h compiles against $REF (syntax only)"
