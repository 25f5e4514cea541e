"""Constants the reference takes from the un-vendored `pem_core.constants` (cathode.py:10, plume.py:12,
thruster.py:31).  Their exact upstream values cannot be read offline, so they
are plain module attributes here and TORR_2_PA is also an explicit argument of every C-ABI call."""

TORR_2_PA = 133.322                      # Pa per Torr, value of the former hallmd.utils constant
AVOGADRO_CONSTANT = 6.02214076e23        # 1/mol (CODATA 2018, exact)
FUNDAMENTAL_CHARGE = 1.602176634e-19     # C     (CODATA 2018, exact)
# g/mol.  pem_core's own table is not in the reference tree; these are the two entries the golden vectors were generated
# with (tests/golden/make_golden.py).  Any other propellant falls back to Xenon with a warning, as thruster.py:168-173 does;
# update this dict from pem_core.constants.MOLECULAR_WEIGHTS when that package is installed.
MOLECULAR_WEIGHTS = {'Xenon': 131.293, 'Krypton': 83.798}


def set_torr_2_pa(value: float) -> None:
    """Use the installed pem_core's value when one is available: set_torr_2_pa(pem_core.constants.TORR_2_PA)."""
    global TORR_2_PA
    TORR_2_PA = float(value)
