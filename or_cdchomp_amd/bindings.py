"""Keyword-argument front end that builds the module's command strings.

Mirrors the reference's Python layer (/root/reference
pythonsrc/orcdchomp/orcdchomp.py:27-219): same function names, same keyword
names, same key order and number formats in the emitted command, so a caller of
``orcdchomp.orcdchomp.bind(mod)`` can switch modules unchanged.  The emitted
strings are the specification of the C++ parser.
"""
import types


def shquot(s):                                   # orcdchomp.py:39-40
    return "'" + s.replace("'", "'\\''") + "'"


def _name(obj):
    return obj.GetName() if hasattr(obj, "GetName") else obj


def _vec(values):
    return shquot(" ".join(str(v) for v in values))


class _Cmd:
    """Accumulates `key value` pairs in call order, skipping None."""

    def __init__(self, verb):
        self.parts = [verb]

    def body(self, key, obj):
        if obj is not None:
            self.parts.append("%s %s" % (key, shquot(_name(obj))))
        return self

    def fmt(self, key, value, spec):
        if value is not None:
            self.parts.append(("%s " + spec) % (key, value))
        return self

    def raw(self, key, value):
        if value is not None:
            self.parts.append("%s %s" % (key, value))
        return self

    def quoted(self, key, value):
        if value is not None:
            self.parts.append("%s %s" % (key, shquot(value)))
        return self

    def vec(self, key, values):
        if values is not None:
            self.parts.append("%s %s" % (key, _vec(values)))
        return self

    def flag(self, key, on):
        if on is not None and on:
            self.parts.append(key)
        return self

    def text(self):
        return " ".join(self.parts)


def computedistancefield(mod, kinbody=None, cube_extent=None, aabb_padding=None,
                         cache_filename=None, require_cache=None, releasegil=False):
    c = (_Cmd("computedistancefield").body("kinbody", kinbody)
         .fmt("cube_extent", cube_extent, "%f").fmt("aabb_padding", aabb_padding, "%f")
         .quoted("cache_filename", cache_filename).flag("require_cache", require_cache))
    return mod.SendCommand(c.text(), releasegil)


def addfield_fromobsarray(mod, kinbody=None, obsarray=None, sizes=None, lengths=None,
                          pose=None, releasegil=False):
    c = (_Cmd("addfield_fromobsarray").body("kinbody", kinbody).raw("obsarray", obsarray)
         .vec("sizes", sizes).vec("lengths", lengths).vec("pose", pose))
    return mod.SendCommand(c.text(), releasegil)


def removefield(mod, kinbody=None, releasegil=False):
    return mod.SendCommand(_Cmd("removefield").body("kinbody", kinbody).text(), releasegil)


def create(mod, robot=None, adofgoal=None, basegoal=None, floating_base=None, lambda_=None,
           starttraj=None, n_points=None,
           con_tsr=None, con_tsrs=None, start_tsr=None, start_cost=None, everyn_tsr=None,
           use_momentum=None, use_hmc=None, hmc_resample_lambda=None, seed=None,
           epsilon=None, epsilon_self=None, obs_factor=None, obs_factor_self=None,
           no_report_cost=None, dat_filename=None, releasegil=False, derivative=None, **kwargs):
    c = _Cmd("create").body("robot", robot).vec("adofgoal", adofgoal).vec("basegoal", basegoal)
    c.flag("floating_base", floating_base).fmt("lambda", lambda_, "%0.04f")
    if starttraj is not None:
        c.quoted("starttraj", starttraj if isinstance(starttraj, str) else starttraj.serialize(0))
    c.fmt("n_points", n_points, "%d")
    for tsr in ([con_tsr] if con_tsr is not None else []) + list(con_tsrs or []):
        c.parts.append("con_tsr '%s' '%s'" % (tsr[0], tsr[1].serialize()))
    c.fmt("derivative", derivative, "%d")
    if start_tsr is not None:
        c.parts.append("start_tsr '%s'" % start_tsr.serialize())
    if start_cost is not None:
        c.parts.append("start_cost '%s'" % (start_cost if isinstance(start_cost, str)
                                            else "%s %s" % (start_cost[0], start_cost[1])))
    if everyn_tsr is not None:
        c.parts.append("everyn_tsr '%s'" % everyn_tsr.serialize())
    (c.flag("use_momentum", use_momentum).flag("use_hmc", use_hmc)
      .fmt("hmc_resample_lambda", hmc_resample_lambda, "%f").fmt("seed", seed, "%d")
      .fmt("epsilon", epsilon, "%f").fmt("epsilon_self", epsilon_self, "%f")
      .fmt("obs_factor", obs_factor, "%f").fmt("obs_factor_self", obs_factor_self, "%f")
      .flag("no_report_cost", no_report_cost).quoted("dat_filename", dat_filename))
    return mod.SendCommand(c.text(), releasegil)


def iterate(mod, run=None, n_iter=None, max_time=None, trajs_fileformstr=None,
            cost=None, releasegil=False):
    c = (_Cmd("iterate").raw("run", run).fmt("n_iter", n_iter, "%d")
         .fmt("max_time", max_time, "%f").quoted("trajs_fileformstr", trajs_fileformstr))
    cost_data = mod.SendCommand(c.text(), releasegil)
    if cost is not None:
        cost[0] = float(cost_data)


def gettraj(mod, run=None, no_collision_check=None, no_collision_exception=None,
            no_collision_details=None, releasegil=False):
    c = (_Cmd("gettraj").raw("run", run).flag("no_collision_check", no_collision_check)
         .flag("no_collision_exception", no_collision_exception)
         .flag("no_collision_details", no_collision_details))
    return mod.SendCommand(c.text(), releasegil)   # the serialized trajectory document


def destroy(mod, run=None, releasegil=False):
    return mod.SendCommand(_Cmd("destroy").raw("run", run).text(), releasegil)


def runchomp(mod,
             n_iter=None, max_time=None, trajs_fileformstr=None, cost=None,            # -> iterate
             no_collision_check=None, no_collision_exception=None, no_collision_details=None,  # -> gettraj
             releasegil=False, **kwargs):                                                # rest -> create
    """create + iterate + gettraj + destroy (orcdchomp.py:204-219)."""
    run = create(mod, releasegil=releasegil, **kwargs)
    try:
        iterate(mod, run=run, n_iter=n_iter, max_time=max_time, trajs_fileformstr=trajs_fileformstr,
                cost=cost, releasegil=releasegil)
        traj = gettraj(mod, run=run, no_collision_check=no_collision_check,
                       no_collision_exception=no_collision_exception,
                       no_collision_details=no_collision_details, releasegil=releasegil)
    finally:
        destroy(mod, run=run, releasegil=releasegil)
    return traj


def bind(mod):                                   # orcdchomp.py:27-37
    for fn in (computedistancefield, addfield_fromobsarray, removefield, create, iterate,
               gettraj, destroy, runchomp):
        setattr(mod, fn.__name__, types.MethodType(fn, mod))
    return mod


def parse_traj(text):
    """waypoints [n_points][dof] of a serialized trajectory document."""
    import re
    import numpy as np
    m = re.search(r'<data count="(\d+)">\s*(.*?)\s*</data>', text, re.S)
    count = int(m.group(1))
    vals = np.array(m.group(2).split(), dtype=np.float64).reshape(count, -1)
    return vals[:, :-1]
