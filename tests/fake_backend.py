"""Call recorder standing in for atomsmm_amd.backend.HipContext in CPU tests of the HOST logic
(program unrolling, force translation, group definitions).  It computes nothing."""
import torch


class RecordingContext:
    def __init__(self, n_atoms, box, device=0, stream=None, rank=0, world=1):
        self.n, self.box, self.rank, self.world = n_atoms, box, rank, world
        self.torch_device = torch.device('cpu')
        self.calls = []
        self.pairs = []
        self.bonded = []
        self.groups = {}
        self.runs = []
        self._next = 0

    def _new(self):
        self._next += 1
        return self._next - 1

    def pair_create(self, desc, q, sigma, eps, excl_pairs=None, skin=-1.0):
        fid = self._new()
        self.pairs.append(dict(id=fid, family=desc.family, flags=desc.flags, sign=desc.sign, rc=desc.rc, rc0=desc.rc0,
                               rs0=desc.rs0, alpha=desc.alpha, rswitch=desc.rswitch, degree=desc.degree,
                               q=q.copy(), sigma=sigma.copy(), eps=eps.copy(), n_excl=0 if excl_pairs is None else len(excl_pairs)))
        return fid

    def pair_stats(self, fid):
        return dict(n_rest_atoms=0, list_kind=0)

    def pair_share_list(self, fid, host_fid):
        self.calls.append(('pair_share_list', fid, host_fid))

    def pair_set_params(self, fid, q, sigma, eps):
        self.calls.append(('pair_set_params', fid, q.copy(), sigma.copy(), eps.copy()))

    def pair_energy_derivative(self, fid, pos, out):
        self.calls.append(('pair_energy_derivative', fid))

    def pair_set_scale(self, fid, value):
        self.calls.append(('pair_set_scale', fid, value))

    def pair_set_lambda(self, fid, value):
        self.calls.append(('pair_set_lambda', fid, value))

    def bonded_create(self):
        fid = self._new()
        self.bonded.append(dict(id=fid, terms=[], sliced=False))
        return fid

    def bonded_add_terms(self, fid, kind, idx, params, periodic=False, desc=None):
        [b for b in self.bonded if b['id'] == fid][0]['terms'].append((kind, len(idx), bool(periodic)))

    def bonded_finalize(self, fid, sliced=False):
        [b for b in self.bonded if b['id'] == fid][0]['sliced'] = sliced

    def bonded_release(self, fid):
        self.calls.append(('bonded_release', fid))
        self.bonded = [b for b in self.bonded if b['id'] != fid]

    def pme_create(self, alpha, grid, q, Kc=138.935456):
        fid = self._new()
        self.pme = getattr(self, 'pme', [])
        self.pme.append(dict(id=fid, alpha=alpha, grid=list(grid), q=q.copy(), sliced=False))
        return fid

    def pme_set_charges(self, fid, q):
        self.calls.append(('pme_set_charges', fid, q.copy()))

    def pme_set_sliced(self, fid, on=True):
        [p for p in self.pme if p['id'] == fid][0]['sliced'] = bool(on)

    def expr_define(self, code, consts, globals_):
        self.exprs = getattr(self, 'exprs', [])
        self.exprs.append((list(code), list(consts), list(globals_)))
        return len(self.exprs) - 1

    def constraints_create(self, pairs, distances, tolerance=1e-5):
        self.constraints = (len(distances), tolerance)

    def bath_define(self, z, kT):
        self.baths = getattr(self, 'baths', [])
        self.baths.append((z, kT))
        return len(self.baths) - 1

    def bath_define_nhl(self, h, z, kT, Q, friction, slot):
        self.baths = getattr(self, 'baths', [])
        self.baths.append(('nhl', h, z, kT, Q, friction, slot))
        return len(self.baths) - 1

    def bath_define_sin(self, h, z, kT, Q2, friction, slot_v2):
        self.baths = getattr(self, 'baths', [])
        self.baths.append(('sin', h, z, kT, Q2, friction, slot_v2))
        return len(self.baths) - 1

    def iso_define(self, on, LkT=0.0, Q1=0.0, slot_v1=-1):
        self.calls.append(('iso_define', bool(on), LkT, Q1, slot_v1))

    def expr_seed(self, seed):
        self.calls.append(('expr_seed', seed))

    def expr_eval(self, code, consts, globals_, seed, counter, dst=None, total=None):
        self.calls.append(('expr_eval', list(code), list(consts), list(globals_), counter, dst is not None, total is not None))
        if total is not None:
            total.fill_(1.0)

    def bind_state(self, x, v, mass):
        pass

    def bind_buffer(self, slot, buf):
        self.calls.append(('bind_buffer', slot))

    def group_define(self, group, slot, force_ids):
        self.groups[group] = (slot, list(force_ids))

    def run_ops(self, ops, repeat=1):
        self.runs.append(([(o.op, o.a, o.b, o.c, o.coef) for o in ops], repeat))

    def run_ops_host_exchanges(self, ops, repeat, exchange):
        """The resumable form (several ranks, collectives made by the host): nothing is exchanged by this recorder."""
        self.run_ops(ops, repeat)

    def force_eval(self, fid, pos, force, accumulate=False, energy=None):
        self.calls.append(('force_eval', fid, accumulate))

    def mvv(self, v, mass, out):
        pass

    def check(self):
        pass

    def synchronize(self):
        pass

    def close(self):
        pass
