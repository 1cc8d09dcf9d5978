#!/bin/bash
# First contact with a Rust toolchain, as ONE command (round-4 review item 4): everything that has only ever existed as files --
# the patch, the shim, the golden dumper, the bit-exact harness, the CPU baseline -- runs here in the order that localises a failure.
#
#   integration/first_contact.sh [--dry-run] [--reference DIR] [--work DIR] [--size 20] [--skip-big]
#
# Needs (real run): cargo + a nightly toolchain (the reference's README: `rustup override set nightly`), the crates of
# integration/DEPS.md (vendored if the box is offline), an MI355X, this repository built (`python -c 'import __graft_entry__ as g; g.build()'`).
# --dry-run verifies every step that needs no cargo -- paths, the patch, feature names, example files, the symbols the shim binds,
# the environment variables it reads, the TimingTree scope names the report tabulates -- and prints the commands of the rest.
# It runs in the CPU test tier (tests/test_integration_files.py::test_first_contact_dry_run).
set -u
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"; REPO="$(dirname "$HERE")"
DRY=0; REF="${P2_REFERENCE:-/root/reference}"; WORK=""; SIZE=20; SKIPBIG=0
while [ $# -gt 0 ]; do case "$1" in
  --dry-run) DRY=1;; --reference) REF="$2"; shift;; --work) WORK="$2"; shift;; --size) SIZE="$2"; shift;; --skip-big) SKIPBIG=1;;
  *) echo "unknown argument $1"; exit 2;; esac; shift; done
[ -n "$WORK" ] || WORK="$(mktemp -d /tmp/p2hot_first_contact.XXXXXX)"
LIBDIR="$REPO/plonky2_amd"; REPORT="$WORK/first_contact_report.md"; FAILS=0
step() { echo; echo "== step $1: $2"; }
ok()   { echo "   ok: $*"; }
bad()  { echo "   FAIL: $*"; FAILS=$((FAILS+1)); }
need() { if [ -e "$1" ]; then ok "$1"; else bad "missing $1"; fi; }
has()  { if grep -q -- "$2" "$1" 2>/dev/null; then ok "$3"; else bad "$3 -- \`$2\` not in $1"; fi; }
run()  { echo "   \$ $*"; if [ $DRY = 1 ]; then return 0; fi; "$@"; local rc=$?; [ $rc = 0 ] || bad "exit code $rc: $*"; return $rc; }

step 0 "inputs"
need "$REF/Cargo.toml"; need "$REF/plonky2/src/fri/oracle.rs"; need "$HERE/plonky2_p2hot.patch"; need "$HERE/p2hot.rs"; need "$HERE/p2hot_dump_goldens.rs"
if [ -e "$LIBDIR/libp2hot.so" ]; then ok "$LIBDIR/libp2hot.so"; else bad "libp2hot.so is not built: python -c 'import __graft_entry__ as g; g.build()'"; fi
if command -v cargo >/dev/null 2>&1; then ok "cargo: $(cargo --version)"; HAVE_CARGO=1; else echo "   (no cargo on this box)"; HAVE_CARGO=0; fi
if [ $DRY = 0 ] && [ $HAVE_CARGO = 0 ]; then echo "cargo is required for a real run; use --dry-run here"; exit 3; fi

step 1 "a scratch copy of the reference with the patch applied ($WORK/plonky2)"
rm -rf "$WORK/plonky2" && mkdir -p "$WORK" && cp -r "$REF" "$WORK/plonky2" || { bad "copy failed"; exit 1; }
( cd "$WORK/plonky2" && rm -rf .git && patch -p1 -s --dry-run < "$HERE/plonky2_p2hot.patch" ) && ok "the patch applies to $REF" || bad "the patch does not apply"
( cd "$WORK/plonky2" && patch -p1 -s < "$HERE/plonky2_p2hot.patch" ) || bad "patch failed"
cmp -s "$WORK/plonky2/plonky2/src/p2hot.rs" "$HERE/p2hot.rs" && ok "plonky2/src/p2hot.rs is integration/p2hot.rs" || bad "the patch's p2hot.rs differs from integration/p2hot.rs (run tools/make_rust_patch.py)"
cmp -s "$WORK/plonky2/plonky2/examples/p2hot_dump_goldens.rs" "$HERE/p2hot_dump_goldens.rs" && ok "the example is integration/p2hot_dump_goldens.rs" || bad "the patch's dumper differs"

step 2 "what cargo will be asked for exists under the names used below"
P="$WORK/plonky2"
has "$P/plonky2/Cargo.toml" '^p2hot = \["std"\]' "feature p2hot of the plonky2 crate"
has "$P/starky/Cargo.toml" 'p2hot = \["std", "plonky2/p2hot"\]' "feature p2hot of the starky crate (forwarded)"
has "$P/plonky2/build.rs" 'P2HOT_LIB_DIR' "build.rs takes the library directory from P2HOT_LIB_DIR"
has "$P/plonky2/src/lib.rs" 'pub mod p2hot;' "the module is declared"
need "$P/plonky2/examples/p2hot_dump_goldens.rs"; need "$P/plonky2/examples/bench_recursion.rs"; need "$P/plonky2/examples/factorial.rs"; need "$P/plonky2/examples/square_root.rs"
grep -q 'autoexamples *= *false' "$P/plonky2/Cargo.toml" && bad "autoexamples is off: the new example needs an [[example]] entry" || ok "examples are auto-discovered (no [[example]] entry needed)"
has "$P/plonky2/examples/bench_recursion.rs" 'default_value="14"' "bench_recursion takes --size (default 14)"
for v in P2HOT_DEVICE P2HOT_LEAVES P2HOT_DISABLE P2HOT_LEAVES_SYNC; do has "$P/plonky2/src/p2hot.rs" "\"$v\"" "the shim reads $v"; done
for s in '"IFFT"' '"FFT + blinding"' '"transpose LDEs"' '"build Merkle tree"'; do has "$P/plonky2/src/fri/oracle.rs" "$s" "TimingTree scope $s (CPU body)"; done
has "$P/plonky2/src/fri/oracle.rs" '"p2hot commit"' 'TimingTree scope "p2hot commit"'; has "$P/plonky2/src/fri/oracle.rs" '"p2hot prove_openings"' 'TimingTree scope "p2hot prove_openings"'
# every extern symbol of the shim is exported by the library that will be linked
if [ -e "$LIBDIR/libp2hot.so" ]; then
  EXT=$(awk '/extern "C" \{/,/^\}/' "$HERE/p2hot.rs" | grep -o 'pub fn p2hot_[a-z0-9_]*' | sed 's/pub fn //' | sort -u)
  MISS=$(comm -23 <(echo "$EXT") <(nm -D --defined-only "$LIBDIR/libp2hot.so" | awk '{print $3}' | sort -u))
  [ -z "$MISS" ] && ok "all $(echo "$EXT" | wc -l) extern \"C\" symbols of the shim are exported by libp2hot.so" || bad "not exported by libp2hot.so: $MISS"
fi
has "$HERE/DEPS.md" 'rayon' "integration/DEPS.md lists the crates to vendor"

export P2HOT_LIB_DIR="$LIBDIR" RUSTFLAGS="${RUSTFLAGS:--Ctarget-cpu=native}"
cd "$P" || exit 1
# the first thing rustc says about 1 500 lines of unsafe FFI it has never seen should be a type error after seconds, not a link
# error after a release build: type-check both crates with the feature on (and once with it off: the patch's cfg-gated hunks must
# leave the default build alone) before anything is compiled.  INTEGRATION.md "Compile-risk ledger" lists what this step judges.
step 2.5 "type check (no codegen): feature on, then off"
run cargo check --features p2hot -p plonky2 -p starky --examples --tests
run cargo check -p plonky2 -p starky
step 3 "build (feature on)";            run cargo build --release --features p2hot -p plonky2 --examples
step 4 "the reference's own bytes for the oracle (no GPU, feature off): tests/golden/reference_run.json"
BIG=""; [ $SKIPBIG = 1 ] || BIG="--big"
run cargo run --release -p plonky2 --example p2hot_dump_goldens -- --out "$WORK/reference_run.json" $BIG
# the oracle is checked against the file where the dumper wrote it (P2_REFERENCE_RUN); only a file that PASSED becomes the
# committed golden.  (No subshell: `run` counts its failures in this shell's FAILS.)
if run env -C "$REPO" P2_REFERENCE_RUN="$WORK/reference_run.json" python -m pytest tests/test_oracle.py -q -k reference_run; then
  run cp "$WORK/reference_run.json" "$REPO/tests/golden/reference_run.json"
else
  echo "   (reference_run.json stays in $WORK: the oracle does not reproduce it)"
fi
step 5 "bit-exact harness: every replaced body, CPU vs GPU, one process (2^12 circuits; then the ignored 2^16 / 2^20 ones)"
run cargo test --release --features p2hot -p plonky2 p2hot:: -- --test-threads=1
[ $SKIPBIG = 1 ] || run cargo test --release --features p2hot -p plonky2 p2hot:: -- --test-threads=1 --ignored
run cargo test --release --features p2hot -p starky
step 6 "C3: bench_recursion --size $SIZE, CPU prover (P2HOT_DISABLE=1) and GPU path, TimingTree scopes side by side"
if [ $DRY = 1 ]; then
  echo "   \$ P2HOT_DISABLE=1 cargo run --release --features p2hot --example bench_recursion -- -vv --size $SIZE > cpu.log"
  echo "   \$ cargo run --release --features p2hot --example bench_recursion -- -vv --size $SIZE > gpu.log"
else
  P2HOT_DISABLE=1 cargo run --release --features p2hot -p plonky2 --example bench_recursion -- -vv --size "$SIZE" > "$WORK/cpu.log" 2>&1 || bad "CPU run failed"
  cargo run --release --features p2hot -p plonky2 --example bench_recursion -- -vv --size "$SIZE" > "$WORK/gpu.log" 2>&1 || bad "GPU run failed"
  { echo "## bench_recursion --size $SIZE on $(hostname): $(nproc) logical CPUs ($(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)), $(date -u +%F)"; echo
    python "$REPO/tools/timing_tree_table.py" "$WORK/cpu.log" "$WORK/gpu.log" --labels "Rust/rayon CPU prover (P2HOT_DISABLE=1),p2hot on the MI355X"; } > "$REPORT"
  cat "$REPORT"; echo "   -> append $REPORT to BASELINE.md section 3"
fi
step 7 "C1 plumbing: the factorial and square_root examples through the shim (square_root asserts data == data_from_bytes)"
run cargo run --release --features p2hot -p plonky2 --example factorial
run cargo run --release --features p2hot -p plonky2 --example square_root
echo; if [ $FAILS = 0 ]; then echo "first contact: all steps ok$([ $DRY = 1 ] && echo ' (dry run: cargo steps printed, not run)')"; else echo "first contact: $FAILS failure(s)"; fi
[ $DRY = 1 ] && rm -rf "$WORK/plonky2"
exit $([ $FAILS = 0 ] && echo 0 || echo 1)
