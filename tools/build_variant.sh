#!/bin/bash
# tools/build_variant.sh <git-rev|WORK> <out.so> [extra hipcc flags]: builds libdnagpu from a revision's csrc (or the working tree) into
# another file, for A/B runs on one box (DNAGPU_LIB_PATH=<out.so>)
set -e
REV=$1; OUT=$(readlink -f $2); shift 2
ROOT=$(git rev-parse --show-toplevel)
TMP=$(mktemp -d)
if [ "$REV" = "WORK" ]; then
  mkdir -p $TMP/dna-sequences-pg-extension_amd $TMP/include
  cp -r $ROOT/dna-sequences-pg-extension_amd/csrc $TMP/dna-sequences-pg-extension_amd/
  cp $ROOT/include/dnagpu.h $TMP/include/
else
  git -C $ROOT archive $REV dna-sequences-pg-extension_amd/csrc include | tar -x -C $TMP
fi
cd $TMP/dna-sequences-pg-extension_amd/csrc
rm -f *.o
make -s OUT=$OUT HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value $*" 2>&1 | grep -E "error" || true
rm -rf $TMP
ls -la $OUT
