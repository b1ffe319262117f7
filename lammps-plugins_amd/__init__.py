"""MI355X-native AEAM / REBO-MoS pair-style hot paths.

The directory name carries a hyphen (it mirrors the upstream repo name), so the package is
registered under the importable name ``lammps_plugins_amd`` by ``__graft_entry__.load_package()``.
"""
