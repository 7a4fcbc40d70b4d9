#!/usr/bin/env bash
# TEST / MEASUREMENT INFRASTRUCTURE -- not part of the product path.
#
# Stages the UNMODIFIED reference package (timsainb/noisereduce, /root/reference/noisereduce) under the git-ignored
# oracle/_ref/ so that it travels to the GPU box with the gpurun snapshot (exactly like the built .so does) and
# bench.py's `cpu_baseline` leg can time the reference's own CPU path ("kind": "reference") on the GPU box's host
# cores.  Nothing under noisereduce_amd/ may import it; only bench.py's cpu_baseline() and tests/ do.
# No reference source enters the repository's history: oracle/_ref/ is listed in .gitignore (and NOT in
# .gpurunignore).  MANIFEST.sha256 records what was staged, so a run can say which files it timed.
#
#   usage: oracle/make_ref.sh [reference-root]      (default /root/reference)
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
src="${1:-/root/reference}"
dst="$here/_ref"
if [ ! -d "$src/noisereduce" ]; then
  echo "make_ref.sh: $src/noisereduce not found (nothing staged)" >&2
  exit 3
fi
# staged into a scratch directory next to the target, then swapped in: concurrent callers (several processes running
# build() at once) never see a half-copied tree, and a failed copy leaves the previous staging in place
final="$dst"
dst="$(mktemp -d "$here/_ref.tmp.XXXXXX")"
trap 'rm -rf "$dst"' EXIT
# python sources only (no __pycache__, no notebooks / assets)
(cd "$src" && find noisereduce -name '*.py' -print0 | sort -z | xargs -0 -I{} cp --parents {} "$dst/")
(cd "$dst" && find noisereduce -name '*.py' -print0 | sort -z | xargs -0 sha256sum > MANIFEST.sha256)
{
  echo "source: $src"
  (cd "$src" && git rev-parse HEAD 2>/dev/null | sed 's/^/reference git HEAD: /') || true
  grep -m1 -E '^\s*version' "$src/setup.py" 2>/dev/null | sed 's/^\s*/setup.py /' || true
  echo "files: $(wc -l < "$dst/MANIFEST.sha256")"
} > "$dst/SOURCE.txt"
n="$(wc -l < "$dst/MANIFEST.sha256")"
old="$(mktemp -d "$here/_ref.old.XXXXXX")"
if [ -d "$final" ]; then mv "$final" "$old/_ref"; fi
mv "$dst" "$final"
rm -rf "$old"
echo "make_ref.sh: staged $n files under $final"
