#!/bin/bash
# Build the C part of the oracle (checker only; never linked into the product).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$here/_build"
gcc -O2 -ffp-contract=off -shared -fPIC "$here/vq_ref.c" -o "$here/_build/libvq_ref.so" -lm
echo "built $here/_build/libvq_ref.so"
