"""tools/gen_record_dev.py — the build step that reads a generated pair {iLQG_problem.h, iLQG_func.c} and tells the device
wrapper (ilqg_kernels.hip, namespace ilqgdev) the members of trajEl_t, the order in which init_running / bp_derivsL assign
derivative entries, and whether limitsU() stores anything but zeros as the limits' gradients.  The pair is compiled
UNMODIFIED either way; what the scan decides is whether the wave-mapped kernels may hand the callbacks a private element
with proxies (ILQG_DEV_RECORDS 1) — so it has to say no whenever it cannot vouch for every access to a derivative member.
CPU only: the scanner on small hand-written pairs in the template's form (iLQG_problem.tem:23-51, iLQG_func.tem:75-119,
262-347) and on the pairs of this repository."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_record_dev as G  # noqa: E402

HEADER = """
#define N_X 2
#define N_U 1
#define sizeofQxx 3
#define sizeofQuu 1
#define sizeofQxu 2
typedef struct {
    double x[N_X];
    double u[N_U];
    double lower[N_U];
    double upper[N_U];
    double lower_sign[N_U];
    double upper_sign[N_U];
    double lower_hx[N_X*N_U];
    double upper_hx[N_X*N_U];
    double l[N_U];
    double L[N_U*N_X];
    double c;
    double cx[N_X];
    double cxx[sizeofQxx];
    double cu[N_U];
    double cuu[sizeofQuu];
    double cxu[sizeofQxu];
    double fx[N_X*N_X];
    double fu[N_X*N_U];
#if FULL_DDP
    double fxx[N_X*sizeofQxx];
    double fuu[N_X*sizeofQuu];
    double fxu[N_X*sizeofQxu];
#endif
    double s;
} trajEl_t;
typedef struct { double x[N_X]; double c; double cx[N_X]; double cxx[sizeofQxx]; } trajFin_t;
"""


def func(derivs, init="", limits_hx="hx_[0]= 0.0;\n                    hx_[1]= 0.0;", extra=""):
    return textwrap.dedent("""
        #include "iLQG.h"
        static int bp_derivsL(trajEl_t *t, int k, double **p) {
            const double *x= t->x;
        %s
            return 1;
        }
        static int bp_derivsF(trajFin_t *t, int k, double **p) {
            t->cx[0]= 1.0;
            t->cxx[0]= 2.0;
            return 1;
        }
        static void limitsU(trajEl_t *t, int k, double **p, int N) {
            int i, lower_idx[N_U];
            double *hx_, *h_sign;
            for(i= 0; i<N_U; i++) { lower_idx[i]= -1; t->lower[i]= -INF; t->upper[i]= INF; }
            hx_= t->lower_hx; h_sign= t->lower_sign;
            switch(lower_idx[0]) {
                case 0:
                    %s
                    h_sign[0]= -1.0;
                    break;
            }
        }
        static int init_running(trajEl_t *t, tOptSet *o) {
            int k;
            for(k= 0; k<o->n_hor; k++, t++) {
        %s
            }
            return 1;
        }
        %s
        """) % (textwrap.indent(derivs, "    "), limits_hx, textwrap.indent(init, "        "), extra)


def scan(tmp_path, derivs, fd=1, **kw):
    (tmp_path / "iLQG_problem.h").write_text(HEADER)
    (tmp_path / "iLQG_func.c").write_text(func(derivs, **kw))
    out = tmp_path / "out.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_record_dev.py"), str(tmp_path), str(fd), str(out)])
    return out.read_text()


def long_run(member, n, guard=True):
    lines = []
    for i in range(n):
        lines.append("t->%s[%d]= x[0]*%d.0;" % (member, i, i + 1))
        if guard:
            lines.append('if(isNANorINF(t->%s[%d])) { PRNT("    @k %%d: %s[%d] in line %%d is nan or inf: %%g\\n", k, __LINE__-1, t->%s[%d]); return 0; }'
                         % (member, i, member, i, member, i))
    return "\n".join(lines)


def test_a_pair_in_the_template_form_is_accepted(tmp_path):
    derivs = long_run("fx", 4) + "\n" + long_run("fu", 2) + "\n#if FULL_DDP\n" + long_run("fxx", 6) + "\n#endif\n" + long_run("cx", 2) + "\n" + long_run("cxx", 3)
    text = scan(tmp_path, derivs, init="t->cuu[0]= 0.0;\nt->cxu[0]= 0.0;\nt->cxu[1]= 0.0;\n#if FULL_DDP\nmemset(t->fuu, 0, sizeof(double)*N_X*sizeofQuu);\n#endif")
    assert "#define ILQG_DEV_RECORDS 1" in text and "#define ILQG_DEV_LIMITS_STATE_FREE 1" in text
    # members in the header's order, derivative arrays as proxies, the auxiliary a scalar
    members = text.split("#define ILQG_DEV_MEMBERS(SCALAR, ARRAY, PROXY)")[1].split("\n")[0]
    assert members.index("ARRAY(x,") < members.index("SCALAR(c)") < members.index("PROXY(cx,") < members.index("PROXY(fxu,") < members.index("SCALAR(s)")
    # the assignment order, memset spelled out, FULL_DDP resolved
    assert "#define ILQG_DEV_SEQ_DERIVS(E) E(fx,0) E(fx,1) E(fx,2) E(fx,3) E(fu,0) E(fu,1) E(fxx,0)" in text and text.rstrip().endswith("E(cx,0) E(cx,1) E(cxx,0) E(cxx,1) E(cxx,2)")
    assert "#define ILQG_DEV_SEQ_INIT(E) E(cuu,0) E(cxu,0) E(cxu,1) E(fuu,0) E(fuu,1)" in text
    in_order = "\n".join(long_run(m, n) for m, n in (("cx", 2), ("cxx", 3), ("cu", 1), ("cuu", 1), ("cxu", 2), ("fx", 4), ("fu", 2)))
    without = scan(tmp_path, in_order + "\n#if FULL_DDP\n" + long_run("fxx", 6) + "\n#endif", fd=0)
    assert "#define ILQG_DEV_RECORDS 1" in without and "fxx" not in without.split("ILQG_DEV_SEQ_DERIVS")[1] and "PROXY(fxx" not in without


@pytest.mark.parametrize("derivs,extra,why", [
    (long_run("fx", 4) + "\nt->fu[0]= t->fx[1]*2.0;\nt->fu[1]= 1.0;", "", "is accessed outside an assignment"),           # an entry read back
    (long_run("fx", 4) + "\n{ int i; for(i= 0; i<2; i++) t->fu[i]= 1.0; }", "", "is accessed outside an assignment"),      # an index that is no literal
    (long_run("fx", 4) + "\n" + long_run("fu", 2), "static void other(trajEl_t *t) { t->cx[0]= 1.0; }", "is accessed outside an assignment"),
    (long_run("fx", 4) + "\nt->fx[0]= 2.0;\n" + long_run("fu", 2), "", "assigns an entry twice"),
    ("t->fx[0]= 1.0;\nt->fu[1]= 1.0;\nt->cxx[2]= 1.0;\nt->fx[3]= 2.0;", "", "too short to pay"),                              # scattered entries
    (long_run("fx", 4) + "\n#ifdef SOMETHING\n" + long_run("fu", 2) + "\n#endif", "", "preprocessor directive"),
])
def test_a_pair_the_scan_cannot_vouch_for_is_built_as_before(tmp_path, derivs, extra, why):
    text = scan(tmp_path, derivs, extra=extra)
    assert "#define ILQG_DEV_RECORDS 0" in text and why in text, text
    assert "ILQG_DEV_MEMBERS" not in text


def test_limits_that_depend_on_the_state_are_seen(tmp_path):
    assert "#define ILQG_DEV_LIMITS_STATE_FREE 0" in scan(tmp_path, long_run("fx", 4), limits_hx="hx_[0]= 2.0*x[1];\n                    hx_[1]= 0.0;")
    assert "#define ILQG_DEV_LIMITS_STATE_FREE 0" in scan(tmp_path, long_run("fx", 4), limits_hx="*hx_++ = 0.0;")           # a form the rule does not know
    assert "#define ILQG_DEV_LIMITS_STATE_FREE 0" in scan(tmp_path, long_run("fx", 4), limits_hx="/* nothing */")            # no gradient assignment at all
    assert "#define ILQG_DEV_LIMITS_STATE_FREE 1" in scan(tmp_path, long_run("fx", 4))


def test_the_pairs_of_this_repository(tmp_path):
    """the hint-free n = 16 pair (emitted on the spot) is accepted with the run structure the kernels count on; the small and the
    hinted pairs keep the old build, each for its reason; hxtest's limits depend on the state"""
    plain = tmp_path / "plain"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_problem.py"), "--plain", os.path.join(ROOT, "problems", "defs", "synth16x8.py"), str(plain)],
                          stdout=subprocess.DEVNULL)
    members, proxied, seqs, (n, runs) = G.scan(str(plain), 1)
    assert proxied == list(G.PROXIED) and n == 5224 and runs <= 100 and len(seqs["init_running"]) == 284
    assert G.limits_state_free(str(plain)) is True
    for problem, fd, reason in (("carparking", 1, "too short to pay"), ("synth16x8", 1, "is accessed outside an assignment")):
        with pytest.raises(G.Unsupported, match=reason):
            G.scan(os.path.join(ROOT, "problems", problem), fd)
    assert G.scan(os.path.join(ROOT, "problems", "synth16p"), 1)[3][0] == 5224
    assert G.limits_state_free(os.path.join(ROOT, "problems", "hxtest")) is False
