"""MI355X-native integrator for the L'Heureux (2018) five-field diagenetic model.

The directory name carries the upstream repository's name and is not a Python identifier; import the
package as ``marlpde_amd`` (the one-line shim ``marlpde_amd.py`` at the repository root points here):

    from marlpde_amd.parameters import Map_Scenario, Solver, Tracker
    from marlpde_amd.Evolve_scenario import integrate_equations

Modules mirror the reference's ``marlpde`` package: ``parameters``, ``LHeureux_model``,
``Evolve_scenario``; ``sweep`` and ``domain`` add the multi-GPU drivers.  All compute goes through
``csrc/libmarl_hip.so`` (C ABI: ``include/marl_hip.h``); nothing here computes on the CPU.
"""
__version__ = "0.1.0"
