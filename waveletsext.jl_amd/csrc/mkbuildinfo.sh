#!/bin/sh
# Writes wx_buildinfo_gen.h (the string wx_build_info() returns) when its content would change:
#   src   = first 12 hex digits of the SHA-256 over every *.hip / *.h of this directory and the public header (what the
#           library was built from, whether or not the tree was committed)
#   git   = HEAD of the enclosing repository, "+dirty" when csrc/ or include/ differ from it; kept from the previous
#           build when there is no repository (the GPU box receives a snapshot without .git and the prebuilt library)
#   hipcc = compiler version, arch and flags
# usage: mkbuildinfo.sh <hipcc> <arch> <flags...>
set -e
cd "$(dirname "$0")"
HIPCC="$1"; ARCH="$2"; shift 2
FLAGS="$*"
SRC=$(cat $(ls *.hip *.h ../../include/waveletsext_hip.h | grep -v wx_buildinfo_gen.h | LC_ALL=C sort) | sha256sum | cut -c1-12)
if git rev-parse --short=12 HEAD >/dev/null 2>&1; then
    GIT=$(git rev-parse --short=12 HEAD)
    if [ -n "$(git status --porcelain -- . ../../include 2>/dev/null)" ]; then GIT="$GIT+dirty"; fi
elif [ -f wx_buildinfo_gen.h ]; then
    GIT=$(sed -n 's/.* git=\([^ ]*\) .*/\1/p' wx_buildinfo_gen.h | head -1)
else
    GIT=unknown
fi
HV=$("$HIPCC" --version 2>/dev/null | sed -n 's/^HIP version: *//p' | head -1)
CV=$("$HIPCC" --version 2>/dev/null | sed -n 's/.*clang version \([^ ]*\).*/\1/p' | head -1)
LINE="libwaveletsext_hip 0.1.0 src=$SRC git=$GIT arch=$ARCH hip=$HV clang=$CV flags=[$FLAGS]"
NEW="#define WX_BUILD_INFO \"$LINE\""
if [ ! -f wx_buildinfo_gen.h ] || [ "$(cat wx_buildinfo_gen.h)" != "$NEW" ]; then
    printf '%s\n' "$NEW" > wx_buildinfo_gen.h
fi
