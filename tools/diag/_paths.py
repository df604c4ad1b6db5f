"""Import paths of the diagnostic scripts (run from anywhere: python tools/diag/<script>.py): the repository root, tests/
(parity helpers, the numpy interpreter) and tests/golden/ (seeded recipes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
