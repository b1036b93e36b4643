"""CPU ORACLE (test infrastructure) for the Pennes bio-heat step the reference obtains from
`BabelViscoFDTD.tools.RayleighAndBHTE.BHTE` (package absent from /root/reference; call sites
ThermalModeling/CalculateTemperatureEffects.py:365-456, 960). PARITY UNPINNED against that package: this is the
documented explicit scheme of csrc/bfd_bhte.hip restated with numpy float32 arrays in the same operation order
(so it agrees with the device to rounding), checked by analytic known-answer tests. Only tests/ may import this."""
import numpy as np


def bhte(T0, dose0, q, mat, cd, cp, Tcore, dt, nSteps, nStepsOn, field_of_step=None):
    """All arrays (N1,N2,N3); cd, cp per material float32; q = increment of one ON step. Returns (T, dose).
    With field_of_step (length nSteps, -1 = no heating) q is (nFields,N1,N2,N3) and nStepsOn is ignored."""
    T = np.array(T0, np.float32)
    dose = np.array(dose0, np.float32)
    cdv = np.asarray(cd, np.float32)[mat][1:-1, 1:-1, 1:-1]
    cpv = np.asarray(cp, np.float32)[mat][1:-1, 1:-1, 1:-1]
    qa = np.asarray(q, np.float32)
    if field_of_step is None:
        qa = qa[None]
        field_of_step = [0 if s < nStepsOn else -1 for s in range(nSteps)]
    Tc = np.float32(Tcore)
    dtm = np.float32(dt / 60.0)
    for s in range(nSteps):
        c = T[1:-1, 1:-1, 1:-1]
        sm = ((((T[:-2, 1:-1, 1:-1] + T[2:, 1:-1, 1:-1]) + T[1:-1, :-2, 1:-1]) + T[1:-1, 2:, 1:-1]) + T[1:-1, 1:-1, :-2]) + T[1:-1, 1:-1, 2:]
        tn = c + cdv * (sm - np.float32(6.0) * c)
        tn = tn + cpv * (Tc - c)
        if field_of_step[s] >= 0:
            tn = tn + qa[field_of_step[s]][1:-1, 1:-1, 1:-1]
        Tn = T.copy()
        Tn[1:-1, 1:-1, 1:-1] = tn
        T = Tn
        R = np.where(T >= np.float32(43.0), np.float32(0.5), np.float32(0.25))
        dose = dose + dtm * np.power(R, np.float32(43.0) - T, dtype=np.float32)
    return T, dose
