"""Hand-made columns for the branch arms the synthetic generator never reaches.

    column 0   previous pH far from the root (5.0): the Newton bracket has to grow (co2calc.F90:920-938)
    column 1   heavy dust over almost no biomass: the QA dust deficit survives a level (BGC_mod.F90:2390-2410)
    column 2   no biomass, no dust: POC%hflux_in == 0 and prod == 0 (BGC_mod.F90:2423)
    column 3   500 m thick levels: P_iron%sflux_out would go negative and is clamped (BGC_mod.F90:2482-2486)
"""
import numpy as np

BIOMASS = ("spC", "spChl", "spFe", "spCaCO3", "diatC", "diatChl", "diatFe", "diatSi", "phaeoC", "phaeoChl", "phaeoFe",
           "diazC", "diazChl", "diazFe", "zooC")


def crafted_columns(synth, nz=60):
    syn = synth.generate(4, nz, col0=20)
    syn["kmax"] = np.array([nz, nz, nz, 12], dtype=np.int32)
    t = syn["tracers"]
    for nm in BIOMASS:
        t[nm][1] *= 1.0e-6
        t[nm][2] = 0.0
    syn["dust"] = syn["dust"].copy()
    syn["dust"][1] = 1.0e-6
    syn["dust"][2] = 0.0
    # column 3: little sinks out of the top level, a bloom with almost no iron in the level below it
    for nm in BIOMASS:
        t[nm][3, 0] *= 1.0e-3
        t[nm][3, 1] = t[nm][3, 0] * 5.0e4
    for nm in ("spFe", "diatFe", "phaeoFe", "diazFe"):
        t[nm][3, 1] *= 1.0e-3
    t["fe"][3, 1] = 1.0e-5
    dz = syn["dz"].copy()
    dz[3] = 5.0e4
    zbot = np.cumsum(dz, axis=1)
    syn["dz"], syn["zbot"], syn["zmid"] = dz, zbot, zbot - 0.5 * dz
    ph_prev = np.zeros((4, nz))
    ph_prev[0] = 5.0
    return syn, ph_prev
