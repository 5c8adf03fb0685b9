"""The build identity (bgs_build_id) belongs to what was COMPILED: a changed compile flag rebuilds every object without a
`make clean`, and the id the library reports is folded from the ids its kernel objects were compiled with.  CPU only
(hipcc cross-compiles gfx950); three full builds and a partial one of the library in a scratch copy of csrc/, about 80 s on 8 cores."""

import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "board-game-simulator-python_amd")

OBJECTS = ["bgs_capi", "bgs_host", "bgs_multi", "bgs_pipeline", "connect_kernels", "bounce_kernels", "generic_kernels"]


def _ids(lib_path):
    """(build id, the three kernel units' ids) of a library file, read in a child process."""
    code = (
        "import ctypes, sys; l = ctypes.CDLL(sys.argv[1]); l.bgs_build_id.restype = ctypes.c_char_p; "
        "l.bgs_kernel_unit_id.restype = ctypes.c_char_p; l.bgs_kernel_unit_id.argtypes = [ctypes.c_int]; "
        "print(l.bgs_build_id().decode(), *[l.bgs_kernel_unit_id(i).decode() for i in range(3)])"
    )
    out = subprocess.check_output([sys.executable, "-c", code, lib_path], text=True).split()
    return out[0], out[1:]


def _make(csrc, *args):
    return subprocess.run(["make", "-C", csrc, "-j8", *args], check=True, capture_output=True, text=True).stdout


def _print_id(csrc, *args):
    return subprocess.check_output(["make", "-s", "--no-print-directory", "-C", csrc, "print-id", *args], text=True).strip()


def test_a_changed_flag_rebuilds_every_object_and_changes_the_id(tmp_path):
    csrc = tmp_path / "pkg" / "csrc"
    shutil.copytree(os.path.join(PKG, "csrc"), csrc, ignore=shutil.ignore_patterns("*.o", "flags.stamp", ".pytest_cache"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    csrc = str(csrc)
    lib = str(tmp_path / "pkg" / "libbgs.so")

    _make(csrc)
    first, first_units = _ids(lib)
    assert len(first) == 16 and first != "unknown" and first == _print_id(csrc)
    assert len(set(first_units)) == 3 and all(len(u) == 16 for u in first_units)
    stamps = {o: os.stat(os.path.join(csrc, o + ".o")).st_mtime_ns for o in OBJECTS}

    # nothing changed: nothing is compiled
    assert "hipcc" not in _make(csrc)
    assert {o: os.stat(os.path.join(csrc, o + ".o")).st_mtime_ns for o in OBJECTS} == stamps

    # a flag changes, no `make clean`: EVERY object is compiled again, and the library says so
    extra = "EXTRA_CXXFLAGS=-DBGS_BUILD_IDENTITY_TEST=1"
    out = _make(csrc, extra)
    for o in OBJECTS:
        assert os.stat(os.path.join(csrc, o + ".o")).st_mtime_ns > stamps[o], f"{o}.o was not rebuilt:\n{out}"
    second, second_units = _ids(lib)
    assert second != first and second == _print_id(csrc, extra)
    assert all(a != b for a, b in zip(first_units, second_units))

    # back to the default flags: again everything, and the first id comes back (the id is a function of source + flags)
    _make(csrc)
    assert _ids(lib) == (first, first_units)

    # one kernel source changes: its unit's id and the library's change, the other units keep theirs
    with open(os.path.join(csrc, "bounce_kernels.hip"), "a") as fh:
        fh.write("\n// build identity test\n")
    _make(csrc)
    third, third_units = _ids(lib)
    assert third not in (first, second) and third == _print_id(csrc)
    assert third_units[0] == first_units[0] and third_units[2] == first_units[2] and third_units[1] != first_units[1]

    # ... and so does an edit to that unit's own header (its geometry record and launch tuning), while the host-side
    # plumbing header every unit includes (bgs_internal.h: the batch object, prototypes) moves no id at all: counters
    # under profiles/ are gated per unit (bench.py), a Bounce-only round must not invalidate the Connect counters
    with open(os.path.join(csrc, "bounce_unit.h"), "a") as fh:
        fh.write("\n// build identity test\n")
    with open(os.path.join(csrc, "bgs_internal.h"), "a") as fh:
        fh.write("\n// build identity test\n")
    _make(csrc)
    fourth, fourth_units = _ids(lib)
    assert fourth_units[0] == first_units[0] and fourth_units[2] == first_units[2]
    assert fourth_units[1] not in (first_units[1], third_units[1])
    units = dict(line.split() for line in subprocess.check_output(
        ["make", "-s", "--no-print-directory", "-C", csrc, "print-unit-ids"], text=True).splitlines())
    assert [units["connect"], units["bounce"], units["generic"]] == fourth_units


def test_the_library_in_the_tree_is_the_one_its_sources_give():
    """What __graft_entry__.build() enforces: the loaded library's id equals `make print-id` of the tree."""
    lib = os.path.join(PKG, "libbgs.so")
    assert os.path.exists(lib), "build the library first (python __graft_entry__.py)"
    handle = ctypes.CDLL(lib)
    handle.bgs_build_id.restype = ctypes.c_char_p
    assert handle.bgs_build_id().decode() == _print_id(os.path.join(PKG, "csrc"))
