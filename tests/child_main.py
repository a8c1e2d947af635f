"""Child-process entry of the tests that run a crash-prone leg in a process of its own (tests/helpers.py: run_child):
python tests/child_main.py <test file> <function> <json kwargs> <result path>.  The function's return value (tensors on the CPU,
numbers, dicts, lists) is written with torch.save; an exception or a fatal signal is the parent's test failure."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> int:
    import faulthandler

    faulthandler.enable()  # a fatal signal (the ROCr fault handler's SIGABRT) dumps every python thread's stack to stderr
    path, func, kwargs, out = sys.argv[1:5]
    for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
        if p not in sys.path:
            sys.path.insert(0, p)
    spec = importlib.util.spec_from_file_location(os.path.splitext(os.path.basename(path))[0], path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    result = getattr(mod, func)(**json.loads(kwargs))
    import torch

    torch.save(result, out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
