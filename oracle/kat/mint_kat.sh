#!/bin/bash
# Regenerates tests/golden/kat_reference.json from the reference's own sources (run in the authoring container).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
bash "$HERE/build_ref.sh"
"$HERE/../_ref/kat_ref" > "$HERE/../../tests/golden/kat_reference.json"
echo "wrote tests/golden/kat_reference.json"
"$HERE/../_ref/kat_float" > "$HERE/../../tests/golden/kat_reference_float.json"
echo "wrote tests/golden/kat_reference_float.json"
