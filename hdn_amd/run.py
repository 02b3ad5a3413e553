"""Run one of the reference's own scripts UNCHANGED on the HIP hot path:

    python -m hdn_amd.run /path/to/HDN/tools/test.py --dataset POT210 --config ... --snapshot ...
    python -m hdn_amd.run /path/to/HDN/tools/demo.py --config ... --snapshot ... --video ...

What it does, in order: (1) puts the script's repository root (the parent of tools/) on sys.path, exactly as running the
script from that root would; (2) builds / loads libhdn_hip.so (no CPU fallback: it raises if the library cannot be had);
(3) hdn_amd.install.install(strict=True): rebinds the reference's hot-path symbols (hdn/core/xcorr.py, Oneline_DLTv1/utils.py,
PreShareFeature, STN_Polar, MultiBAN / MultiCircBAN.forward, ModelBuilder.track_proj: the sites are listed in
hdn_amd/install.py); (4) runs the script with runpy under __name__ == "__main__" and its own argv.  The script file and the
reference tree are not modified (tools/test.py:65-72,130,153 and tools/demo.py:168,172 then call the rebound symbols).

Options (before the script path):
    --reference-root DIR   repository root, if it is not the parent of the script's directory
    --preload MOD[:FUNC]   import MOD (and call FUNC()) before anything of the reference is imported, e.g. a site module that
                           provides optional dependencies; the parity tests use it to stub cv2 in the build container
    --no-build             do not (re)compile the library even if its sources are newer
    --no-strict            skip rebinding sites that cannot be imported instead of failing
    --host-tracker         keep the reference's host-side tracker loop (numpy / OpenCV crops, six syncs per frame); by default
                           TRACKS['hdnTrackerHomoProje2e'] is the device-resident loop (hdn_amd.tracker.DeviceTrackerHomo)
"""
from __future__ import annotations

import importlib
import os
import runpy
import sys


def _usage(msg=None):
    if msg:
        print("hdn_amd.run: " + msg, file=sys.stderr)
    print(__doc__, file=sys.stderr)
    raise SystemExit(2)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    root, preload, build, strict, dev_tracker = None, [], True, True, True
    while argv and argv[0].startswith("--"):
        opt = argv.pop(0)
        if opt == "--reference-root":
            root = argv.pop(0) if argv else _usage("--reference-root needs a directory")
        elif opt == "--preload":
            preload.append(argv.pop(0) if argv else _usage("--preload needs MOD[:FUNC]"))
        elif opt == "--no-build":
            build = False
        elif opt == "--no-strict":
            strict = False
        elif opt == "--host-tracker":
            dev_tracker = False
        elif opt in ("-h", "--help"):
            _usage()
        else:
            _usage(f"unknown option {opt}")
    if not argv:
        _usage("no script given")
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        _usage(f"{script} is not a file")
    root = os.path.abspath(root) if root else os.path.dirname(os.path.dirname(script))
    for p in (root, os.path.dirname(script)):  # `python tools/test.py` from the root sees both
        if p not in sys.path:
            sys.path.insert(0, p)

    for spec in preload:
        mod, _, fn = spec.partition(":")
        m = importlib.import_module(mod)
        if fn:
            getattr(m, fn)()

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if build:
        if repo not in sys.path:
            sys.path.append(repo)
        try:
            import __graft_entry__ as entry  # the in-tree build recipe (hipcc --offload-arch=gfx950)
            entry.build()
        except ImportError:
            pass  # installed without the build script: the library must already be there
    from hdn_amd import _lib, install

    _lib.load()  # fail now, loudly, rather than at the first frame
    done = install.install(strict=strict, tracker=dev_tracker)
    print(f"[hdn_amd.run] {len(done)} hot-path sites rebound to libhdn_hip.so; running {script}", file=sys.stderr, flush=True)

    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
