"""
Extract the reference's own known answers for the optimisation half of the hot path and
write them to tests/golden/known_answers.json.  Run in the build container (needs
/root/reference; the GPU box does not have it -- the JSON is what travels).

The reference ships no tests; the only numbers of its oracle it contains are hard-coded in
``lib/post_process.py`` (``tree_depth_complexity`` / ``leaf_complexity``): for its cwh_z
runs the absolute-error tolerances ``err_abs`` (by lib/examples.py:42-45 these ARE
max_v P_theta(abs_frac * v).J over the vertices of the partitioned box), the relative
tolerances ``err_rel``, the tree depths ``tau`` and the leaf counts ``leafs``.  The job
parameters of the first two entries are pinned by ``make_jobs.sh`` (EXAMPLE, MPC_N,
ABS_FRACS, REL_ERRS); the abs_frac of the later entries is not recorded anywhere in the
tree, so only the first two are usable as known-answer tests.
"""

import ast
import json
import os
import re

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'known_answers.json')


def literal_list(src, name, func):
    body = src[src.index('def %s(' % func):]
    m = re.search(r'%s\s*=\s*(\[[^\]]*\])' % name, body)
    return ast.literal_eval(m.group(1))


def shell_array(src, name):
    m = re.search(r'^%s=\(([^)]*)\)' % name, src, re.M)
    return [float(v) for v in m.group(1).split()]


def main():
    pp = open(os.path.join(REF, 'lib', 'post_process.py')).read()
    mj = open(os.path.join(REF, 'make_jobs.sh')).read()
    err_abs = literal_list(pp, 'err_abs', 'tree_depth_complexity')
    err_rel = literal_list(pp, 'err_rel', 'tree_depth_complexity')
    tau = literal_list(pp, 'tau', 'tree_depth_complexity')
    leafs = literal_list(pp, 'leafs', 'leaf_complexity')
    assert err_abs == literal_list(pp, 'err_abs', 'leaf_complexity')
    example = re.search(r'^EXAMPLE=(\w+)', mj, re.M).group(1)
    N = int(re.search(r'^MPC_N=(\d+)', mj, re.M).group(1))
    abs_fracs = shell_array(mj, 'ABS_FRACS')
    rel_errs = shell_array(mj, 'REL_ERRS')
    assert rel_errs == err_rel[:len(rel_errs)]
    runs = []
    for k in range(len(err_abs)):
        runs.append(dict(example=example, N=N, rel_err=err_rel[k], eps_a=err_abs[k],
                         abs_frac=abs_fracs[k] if k < len(abs_fracs) else None,
                         tree_depth=tau[k], leaves=leafs[k]))
    json.dump(dict(source=dict(eps_a='lib/post_process.py:484-485 (err_abs)',
                               rel_err='lib/post_process.py:486 (err_rel)',
                               tree_depth='lib/post_process.py:489 (tau)',
                               leaves='lib/post_process.py:526 (leafs)',
                               job='make_jobs.sh:60-66 (EXAMPLE, MPC_N, ABS_FRACS, REL_ERRS)'),
                   runs=runs), open(OUT, 'w'), indent=1)
    print(open(OUT).read())


if __name__ == '__main__':
    main()
