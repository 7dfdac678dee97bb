#!/bin/sh
# Regenerates tests/golden/ from the reference mount (data files only; no reference source is copied).
set -e
REF=${1:-/root/reference}
OUT=$(dirname "$0")/../tests/golden
for n in chacha20 aes128 aes256; do
  xz -9e -k -c "$REF/resources/gnark/r1cs.$n" > "$OUT/r1cs.$n.xz"      # sha256 pinned in gnark/libraries/prover/impl/library.go:38,43,48
done
for n in chacha20 aes128 aes256; do cp "$REF/gnark/libraries/verifier/impl/generated/vk.$n" "$OUT/"; done
# aux_test_request.json: the JSON literal of gnark/libraries/aux_test.go:150-185, re-typed as data (see tests/golden/README.md)
