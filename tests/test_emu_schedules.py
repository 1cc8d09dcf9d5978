"""The stream-heavy paths under OTHER legal schedules of the emulated runtime.

The emulator's streams are queues (tests/emu/hip_emu.h): by default queued work runs as late as is legal, which is what every
test of the tier runs under.  Here the paths with side streams, event hand-overs and collectives -- the host-pointer commit in
column blocks, the batched entry points, the group commit and opening proof on distinct devices -- run again with random legal
steps mixed in (P2HOT_EMU_ASYNC=random:<seed>): an ordering the program does not enforce
(an overwrite in front of a lagging reader, a host buffer reused under a queued copy) gets several chances to show."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

SELECTION = ["tests/test_parity.py", "tests/test_prove_openings.py", "tests/test_emu_devices.py", "tests/test_async_leaves.py", "-k",
             "(emu and (host_commit_in_blocks or host_pointer_commit_abi or commit_many_equals or prove_openings_many or async_and_natural_leaves)) or "
             "group_commit_on_distinct_devices or group_prove_openings_on_distinct or peer_copy_transports"]


@pytest.mark.parametrize("mode", ["random:11"])   # (the whole tier was run under random:1, random:2 and 0 when the model was built)
def test_stream_heavy_paths_under_another_schedule(mode):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + SELECTION,
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env={**os.environ, "P2HOT_EMU_ASYNC": mode})
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, r.stdout[-3000:] + r.stderr[-2000:]


def test_kernels_with_threads_and_blocks_visited_in_reverse():
    """between two barriers the emulator runs a block's threads one after the other: ascending by default, DESCENDING here
    (P2HOT_EMU_THREADS=reverse, blocks too).  A read of shared or global memory that another thread writes in the same
    barrier interval sees the write in one order and not in the other, so a missing __syncthreads fails one of the two runs.
    (The whole tier was run this way when the switch was added; this keeps the kernels' own parity tests under it.)"""
    sel = ["tests/test_parity.py", "-k", "emu and (fft_ifft or coset_lde or merkle_tree_vs or polynomial_batch_vs or fri_committed or transpose or reverse_index)"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + sel,
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env={**os.environ, "P2HOT_EMU_THREADS": "reverse"})
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, r.stdout[-3000:] + r.stderr[-2000:]
