"""CPU oracle for the rasterization hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the product package (`easy_gaussian_splatting_amd/`) may import this
package.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` use it, and only as the checker.

PARITY UNPINNED: the arithmetic of the reference's hot path lives in the
third-party dependency `gsplat` (requirements.txt:1 `gsplat>=1.0.0`, README.md:16
pins `gsplat==1.0.0`), which is neither vendored under /root/reference nor
installable here.  The reference itself holds no tests or golden vectors for the
`rasterization()` boundary.  This oracle therefore restates gsplat 1.0.0's
published algorithm and is anchored on the reference's own
call site (`model/gaussian.py:353-372`, `:188-197`) plus the two in-tree
conventions that *can* be executed here (`model/utils.py:14-16` SH0 constant,
`model/utils.py:31-55` wxyz quaternion -> rotation), captured as fixtures under
`tests/golden/`.
"""
