#!/bin/bash
# Builds the CTU kernel of a git revision (default HEAD) as kvazaar_amd/lib/variants/libkvz_hip_<name>.so for same-box A/B runs.
# usage: tools/build_ref_variant.sh <name> [rev]
set -e
cd "$(dirname "$0")/.."
name=$1; rev=${2:-HEAD}
d=$(mktemp -d)
mkdir -p $d/kvazaar_amd/csrc $d/include kvazaar_amd/lib/variants
for f in $(git ls-tree --name-only $rev kvazaar_amd/csrc/); do git show $rev:$f > $d/$f; done
for f in $(git ls-tree --name-only $rev include/); do git show $rev:$f > $d/$f; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $EXTRA -o kvazaar_amd/lib/variants/libkvz_hip_$name.so $d/kvazaar_amd/csrc/kvz_hip.hip 2>/dev/null
rm -rf $d
ls -la kvazaar_amd/lib/variants/libkvz_hip_$name.so
