"""CPU: the control flow of syn_bcast_constants (synergynet_amd/csrc/bcast_protocol.h -- the very header the HIP library compiles) on a fake
transport whose collectives time out when a rank is missing (tests/bcast_protocol_harness.cpp, built here with g++).

VERDICT r4 #5 / ADVICE r4: the round-4 function let a root that had loaded nothing (or could not allocate) return BEFORE the first
ncclBroadcast -- every other rank was already inside it and hung.  A collective must fail collectively: every rank returns the same error,
nobody is left inside a collective.  Reference construct replaced: benchmark.py:112 / main_train.py:176 (every rank loads the files itself)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('bcast') / 'harness')
    subprocess.run(['g++', '-O1', '-std=c++17', '-pthread', '-o', exe, os.path.join(ROOT, 'tests', 'bcast_protocol_harness.cpp')], check=True)
    return exe


def run(exe, world, scenario):
    r = subprocess.run([exe, str(world), scenario], capture_output=True, text=True, timeout=60)
    ranks = [re.match(r'rank (\d+) rc (-?\d+) imported (\d) where ?(.*)', l) for l in r.stdout.strip().splitlines()]
    assert all(ranks) and len(ranks) == world, r.stdout
    return r.returncode, [(int(m.group(2)), int(m.group(3)), m.group(4)) for m in ranks]


@pytest.mark.parametrize('world', [2, 3])
def test_healthy_broadcast_reaches_every_rank(harness, world):
    code, ranks = run(harness, world, 'ok')
    assert code == 0 and all(rc == 0 for rc, _, _ in ranks)
    assert sum(imp for _, imp, _ in ranks) == world - 1            # everybody but the root imported the root's bytes


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('scenario,want', [('root_empty', -3), ('root_alloc', -2), ('root_export', -7), ('peer_alloc', -2), ('peer_import', -8)])
def test_a_failure_on_one_rank_is_returned_by_every_rank_and_nobody_hangs(harness, world, scenario, want):
    code, ranks = run(harness, world, scenario)
    assert code == 0, 'a collective timed out: some rank left the protocol between two collectives'
    assert [rc for rc, _, _ in ranks] == [want] * world, ranks
    if scenario.startswith('root') or scenario == 'peer_alloc':
        assert not any(imp for _, imp, _ in ranks)                 # nothing was imported anywhere


def test_the_harness_does_see_a_rank_that_leaves_early(harness):
    """Negative control: without the agreement step a peer that cannot stage the blob returns alone and the others wait in the blob's
    broadcast until the fake transport's timeout -- the defect class this test file exists for."""
    code, ranks = run(harness, 3, 'no_allreduce_peer_alloc')
    assert code == 3
