"""GPU test (-m gpu): randomised water systems -- box shape, boundary mask, slabs / droplets / sparse beads,
random placement in the box -- against the CPU oracle (tools/fuzz_parity.py): step-0 forces, energy, virial and
a 25-step trajectory across one list rebuild.  Thin and empty tiles, open faces and beads handed over outside
the box are where index arithmetic goes wrong first."""
import os
import sys
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_water_systems_match_the_oracle(seed):
    from fuzz_parity import run_cases
    worst, worst_t, bad = run_cases(8, seed, verbose=False)
    assert bad == 0 and worst < 1e-9 and worst_t < 1e-6


def test_random_lipid_tilings_shifts_and_domain_grids_match_the_oracle():
    """the molecular path (exclusions, every bonded kind, charges, Berendsen) on random tilings of the lipid deck,
    rigidly shifted by random vectors, on random grids of emulated domains (tools/fuzz_lipid.py)"""
    from fuzz_lipid import run_cases
    worst, worst_t, bad = run_cases(6, 5, verbose=False)
    assert bad == 0 and worst < 1e-9 and worst_t < 1e-6


def test_feature_combinations_match_the_oracle():
    """thermostat kind (FREE / BERENDSEN / LANGEVIN) x velocity constraints x barostat x RESTRAINT potential on the
    lipid deck: all 24 combinations follow the oracle over 25 steps with mixed step batching (tools/fuzz_features.py)"""
    from fuzz_features import run
    worst, bad = run(verbose=False)
    assert bad == 0 and worst < 1e-6


def test_random_water_systems_on_random_domain_grids():
    """the same random systems on grids of up to 3x3x3 emulated domains: narrow domains (down to one list radius),
    domains that hold no bead at all, beads crossing faces"""
    from fuzz_parity import run_cases
    worst, worst_t, bad = run_cases(10, 21, verbose=False, domains=True)
    assert bad == 0 and worst < 1e-9 and worst_t < 1e-6


def test_one_context_reused_for_different_systems():
    """new box + new beads uploaded into the same context, sizes growing and shrinking: every system matches the oracle
    as on a fresh context (tools/fuzz_reuse.py)"""
    from fuzz_reuse import run
    worst, bad = run(8, 9, verbose=False)
    assert bad == 0
