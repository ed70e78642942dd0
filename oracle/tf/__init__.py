"""CPU restatement of the ``torchfilter`` API subset the reference is written against.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``torchfilter`` is the reference's
third-party dependency, pinned only as ``tarball/master``
(``/root/reference/setup.py:12-15``) and absent from ``/root/reference``; its published
algorithm is restated here, anchored on the reference's own call sites.  No reference
test or fixture exists for it => PARITY UNPINNED for the recursion (known-answer tests
stand in; see ``oracle/__init__.py``).
"""
from . import base, data, filters, train, types  # noqa: F401
