"""
Progress and metrics of a partition run in the reference's formats:
the per-process status record of ``WorkerStatusPublisher`` (lib/worker.py:19-116), the
volume-filling-rate estimator and ETA (the role of lib/scheduler.py:59-152)
and the ``status.txt`` / ``statistics.pkl`` writer of ``MainStatusPublisher``
(lib/scheduler.py:154-362), so that an operator's ``watch cat status.txt`` and the
reference's ``PostProcessor.progress`` (lib/post_process.py:57-70, 144-175) keep working.

What changes is where the numbers come from: a "process" is one GPU rank, and instead of one
MPI message per closed leaf the rank reads the device's counters between frontier sweeps
(``ehm_partition_progress``: pool size, closed-leaf volume by a reduction kernel).
"""

import pickle
import time

import numpy as np


class ForgettingMean:
    """
    Exponentially weighted mean of a sampled scalar: sample i counts with weight rho^(age in
    samples), rho = exp(-call_period / time_constant).  Kept as the two running sums
    (numerator, weight), so there is no start-up case: value = sum rho^a m / sum rho^a.  (The
    recursive estimator of lib/scheduler.py:113-152 converges to the same numbers.)
    """

    def __init__(self, call_period, time_constant):
        self.rho = float(np.exp(-call_period / time_constant))
        self.reset()

    def reset(self):
        self._num = self._den = 0.

    def update(self, sample):
        self._num = sample + self.rho * self._num
        self._den = 1. + self.rho * self._den

    @property
    def value(self):
        return self._num / self._den if self._den else None


class EtaEstimate:
    """Seconds left = unfilled volume fraction / smoothed filling rate (the role of
    lib/scheduler.py:59-111); None until a positive rate has been seen."""

    def __init__(self, call_period, time_constant):
        self.rate = ForgettingMean(call_period, time_constant)

    def update(self, measured_rate):
        self.rate.update(measured_rate)

    def reset(self):
        self.rate.reset()

    def eta(self, volume_filled):
        r = self.rate.value
        return (1. - volume_filled) / r if r else None


class WorkerStatus:
    """
    One rank's status record: the keys of ``WorkerStatusPublisher.data`` (lib/worker.py:21-33)
    with the same meaning.  ``absorb`` replaces the reference's per-leaf increments by the
    device's running totals.
    """

    def __init__(self, algorithm='', clock=time.time):
        self._clock = clock
        self.data = dict(status='idle', current_branch='', current_location='',
                         algorithm=algorithm, volume_filled_total=0.,
                         volume_filled_current=0., simplex_count_total=0,
                         simplex_count_current=0, time_active_total=0.,
                         time_active_current=0., time_idle=0., time_ecc=0., time_lcss=0.)
        self.time_previous = clock()
        self.volume_current = None
        self.extra = {}

    def _tick(self):
        dt = self._clock() - self.time_previous
        self.time_previous += dt
        if self.data['status'] == 'active':
            self.data['time_active_total'] += dt
            self.data['time_active_current'] += dt
        else:
            self.data['time_idle'] += dt
        if self.data['algorithm'] == 'ecc':
            self.data['time_ecc'] += dt
        else:
            self.data['time_lcss'] += dt

    def set_total_volume(self, volume):
        """Volume of the set this rank partitions (the reference's "current root simplex")."""
        self.volume_current = float(volume)

    def update(self, active=None, failed=False, location=None, algorithm=None):
        self._tick()
        if active is not None:
            self.data['status'] = 'active' if active else 'idle'
        if failed:
            self.data['status'] = 'failed'
        if location is not None:
            self.data['current_location'] = location
        if algorithm is not None:
            self.data['algorithm'] = algorithm

    def absorb(self, progress):
        """progress: dict of ``PartitionRun.progress()``."""
        self._tick()
        self.data['volume_filled_total'] = float(progress['volume_closed'])
        if self.volume_current:
            self.data['volume_filled_current'] = \
                float(progress['volume_closed']) / self.volume_current
        # the reference counts +1 per SPLIT (lib/worker.py:274,327,107-109), not per node
        splits = int(progress.get('n_splits', 0))
        self.data['simplex_count_total'] = splits
        self.data['simplex_count_current'] = splits
        self.data['current_location'] = 'depth %d, frontier %d' % (progress['depth'],
                                                                   progress['frontier'])
        self.extra = dict(lp_solves=int(progress['lp_solves']),
                          ipm_iters=int(progress['ipm_iters']), sweeps=int(progress['sweeps']),
                          n_closed=int(progress['n_closed']))


class MainStatusPublisher:
    """
    Writes ``status_file`` (text, same lines as lib/scheduler.py:312-361) and appends
    ``dict(overall=..., process=...)`` records to ``statistics_file`` (pickle stream, same keys
    as lib/scheduler.py:296-309), each with its own period.
    """

    def __init__(self, total_volume, status_file='status.txt', statistics_file='statistics.pkl',
                 status_write_period=0., statistics_save_period=0., clock=time.time,
                 eta_window_duration=10., eta_time_constant=180.):
        self.total_volume = float(total_volume)
        self.status_file = status_file
        self.statistics_file = statistics_file
        self._clock = clock
        self.time_previous, self.time_total = None, 0.
        for path in (status_file, statistics_file):     # blank files, like the reference
            if path:
                open(path, 'w').close()
        self.eta_estimator = EtaEstimate(eta_window_duration, eta_time_constant)
        self.eta_last_measurement = None
        self.write_time_prev = dict(eta=None, status=None, statistics=None)
        self.write_period = dict(eta=eta_window_duration, status=status_write_period,
                                 statistics=statistics_save_period)
        self.last_overall = None

    def update_time(self):
        now = self._clock()
        if self.time_previous is None:
            self.time_previous = now
        dt = now - self.time_previous
        self.time_previous += dt
        self.time_total += dt

    def reset_estimators(self):
        self.eta_estimator.reset()
        self.eta_last_measurement = None
        self.write_time_prev['eta'] = None

    def _due(self, what):
        prev = self.write_time_prev[what]
        return prev is None or self.time_total - prev >= self.write_period[what]

    def update(self, proc_status, num_tasks_in_queue=0, looprates=None, force=False):
        """proc_status: list of ``WorkerStatus.data`` dicts (None for a rank not heard of)."""
        self.update_time()
        looprates = looprates or dict(publisher=0., dispatcher=0., collector=0.)
        save_statistics, write_status, update_eta = (self._due('statistics'),
                                                     self._due('status'), self._due('eta'))
        for what, due in (('statistics', save_statistics), ('status', write_status),
                          ('eta', update_eta)):
            if due:
                self.write_time_prev[what] = self.time_total
        known = [d for d in proc_status if d is not None]
        volume_filled_total = sum(d['volume_filled_total'] for d in known)
        volume_filled_frac = volume_filled_total / self.total_volume
        if update_eta:
            new = dict(t=self.time_total, v=volume_filled_frac)
            last = self.eta_last_measurement
            if last is not None and new['t'] > last['t']:
                self.eta_estimator.update((new['v'] - last['v']) / (new['t'] - last['t']))
            self.eta_last_measurement = new
        if not (force or save_statistics or write_status):
            return None
        algorithms = [d['algorithm'] if d is not None and d['status'] == 'active' else None
                      for d in proc_status]
        eta = self.eta_estimator.eta(volume_filled_frac)
        overall = dict(num_proc_active=sum(d['status'] == 'active' for d in known),
                       num_proc_failed=sum(d['status'] == 'failed' for d in known),
                       num_tasks_in_queue=int(num_tasks_in_queue),
                       volume_filled_total=volume_filled_total,
                       volume_filled_frac=volume_filled_frac,
                       simplex_count_total=sum(d['simplex_count_total'] for d in known),
                       time_elapsed=self.time_total, algorithms_running=algorithms,
                       time_active_total=sum(d['time_active_total'] for d in known),
                       time_idle_total=sum(d['time_idle'] for d in known),
                       eta=eta, scheduler_looprate=looprates)
        self.last_overall = overall
        if (force or save_statistics) and self.statistics_file:
            with open(self.statistics_file, 'ab') as f:
                pickle.dump(dict(overall=overall, process=[None if d is None else dict(d)
                                                           for d in proc_status]), f)
        if (force or write_status) and self.status_file:
            lines = ['# overall',
                     'number of processes active: %d' % overall['num_proc_active'],
                     'number of processes failed: %d' % overall['num_proc_failed'],
                     'number of tasks queue: %d' % overall['num_tasks_in_queue'],
                     'volume filled (total [%%]): %.4e' % (volume_filled_frac * 100.),
                     'simplex_count: %d' % overall['simplex_count_total'],
                     'time elapsed [s]: %d' % self.time_total,
                     'time active (total for all processes [s]): %.0f' %
                     overall['time_active_total'],
                     'time idle (total for all processes [s]): %.0f' %
                     overall['time_idle_total'],
                     'processes: %d x ecc, %d x lcss' % (sum(a == 'ecc' for a in algorithms),
                                                         sum(a == 'lcss' for a in algorithms)),
                     'ETA [s]: %s' % str(eta if eta is None else int(np.round(eta))),
                     'Scheduler loop frequencies [pub,disp,col] [Hz]: [%.2e,%.2e,%.2e]' %
                     (looprates['publisher'], looprates['dispatcher'], looprates['collector'])]
            text = '\n'.join(lines) + '\n\n'
            for i, d in enumerate(proc_status):
                if d is None:
                    continue
                text += '\n'.join([
                    '# proc %d' % i,
                    'status: %s' % d['status'],
                    'algorithm: %s' % d['algorithm'],
                    'current branch:   %s' % d['current_branch'],
                    'current location: %s' % d['current_location'],
                    'volume filled (total [-]): %.4e' % d['volume_filled_total'],
                    'volume filled (current [%%]): %.4e' % (d['volume_filled_current'] * 100.),
                    'simplex count (total [-]): %d' % d['simplex_count_total'],
                    'simplex count (current [-]): %d' % d['simplex_count_current'],
                    'time active (total [s]): %d' % d['time_active_total'],
                    'time active (current [s]): %d' % d['time_active_current'],
                    'time idle (total [s]): %d' % d['time_idle'],
                    'time running ecc (total [s]): %d' % d['time_ecc'],
                    'time running lcss (total [s]): %d' % d['time_lcss']]) + '\n\n'
            with open(self.status_file, 'w') as f:
                f.write(text)
        return overall


def load_statistics(path):
    """
    The reader of lib/post_process.py:57-70: lists of every ``overall`` value over time, keyed
    like the reference's ``PostProcessor.statistics``.
    """
    stats = {}
    with open(path, 'rb') as f:
        while True:
            try:
                data = pickle.load(f)
            except EOFError:
                break
            for key, value in data['overall'].items():
                stats.setdefault(key, []).append(value)
    return stats
