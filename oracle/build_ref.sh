#!/bin/sh
# oracle/build_ref.sh -- TEST INFRASTRUCTURE ONLY.
# Compiles the reference's own C engine from the sources where they lie under
# /root/reference (nothing is copied) into oracle/_ref/ (git-ignored, travels
# with gpurun).  Only runs where /root/reference exists.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${KVARQ_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
[ -f "$REF/csrc/workhorse.c" ] || { echo "build_ref: $REF/csrc/workhorse.c not found, skipping"; exit 0; }
mkdir -p "$OUT/kvarq"
# three-file stub package the reference module imports at init time
# (workhorse.c:1595-1609); these are this repo's files, not reference code
printf "VERSION='oracle'\n" > "$OUT/kvarq/__init__.py"
printf "class FastqFileFormatException(Exception):\n    pass\n" > "$OUT/kvarq/fastq.py"
printf "import logging\nlo = logging.getLogger('kvarq')\n" > "$OUT/kvarq/log.py"
PYINC="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
gcc -O2 -shared -fPIC -pthread -w -include "$HERE/py2compat.h" -I"$PYINC" \
    "$REF/csrc/workhorse.c" -o "$OUT/kvarq/engine.so"
echo "build_ref: built $OUT/kvarq/engine.so"
