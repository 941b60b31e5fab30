"""Text output of the two command lines (scripts/pyHICCUPS:200-210, scripts/pyBHFDR:169-176) and their
argument parsers; the executables in scripts/ are thin wrappers around `main_hiccups` / `main_bhfdr`."""

HICCUPS_FMT = ('{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7:.3g}\t{8}\t{9}\t{10:.3g}\t{11:.3g}\t{12:.3g}'
               '\t{13:.3g}\t{14:.3g}\t{15:.3g}\n')
BHFDR_FMT = '{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7:.3g}\t{8}\t{9}\t{10:.3g}\t{11:.3g}\t{12:.3g}\n'


def _format(fmt, chrom, table, res, sort):
    c = 'chr' + chrom.lstrip('chr')
    keys = sorted(table) if sort else list(table)
    out = []
    for px in keys:
        tmp = table[px]
        out.append(fmt.format(*((c, px[0], px[0] + res, c, px[1], px[1] + res, '.', tmp[3], '.', '.') + tuple(tmp[4:]))))
    return ''.join(out)


def format_hiccups(chrom, table, res, sort=False):
    """16 columns: chrom x x+res chrom y y+res . O . . fold_K p_K q_K fold_Y p_Y q_Y (pyHICCUPS:202-205)."""
    return _format(HICCUPS_FMT, chrom, table, res, sort)


def format_bhfdr(chrom, table, res, sort=False):
    """13 columns (pyBHFDR:171)."""
    return _format(BHFDR_FMT, chrom, table, res, sort)


# ----------------------------------------------------------------------------- command lines
import argparse
import itertools
import logging
import logging.handlers
import os
import sys
import time


def _add_deterministic(group):
    group.add_argument('--deterministic', action='store_true',
                       help='The default since round 5, kept as a flag: the output does not depend on the order of the chromosomes, on how '
                            'they are batched or on --nproc / the number of GPUs (like the reference, whose result is independent of its '
                            'map order, scripts/pyHICCUPS:192-210).  A chromosome\'s tiles are laid out for the width at which its OWN '
                            'widening stops; one that ran under another layout - taken over from the chromosomes before it - is computed '
                            'once more.')
    group.add_argument('--history-dependent', action='store_true',
                       help='No second pass: a chromosome keeps the tile layout it inherited from the '
                            'chromosomes scored before it, and its E / p / q values can differ in the 14th digit between runs that order or '
                            'batch the chromosomes differently (coordinates and counts never do).  The fastest mode where the GPU is the '
                            'bottleneck.')


def _hiccups_parser():
    """Flags of scripts/pyHICCUPS:12-81 (same names and defaults) plus --device / --balanced-on-host."""
    from . import __version__, __reference__
    p = argparse.ArgumentParser(usage='%(prog)s <-O output> [options]',
                                description='HiCCUPS peak calling on AMD MI355X (drop-in for pyHICCUPS of %s).' % __reference__,
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-v', '--version', action='version', version=' '.join(['%(prog)s', __version__]))
    p.add_argument('-O', '--output', help='Output file name.')
    p.add_argument('--logFile', default='pyHICCUPS.log', help='Logging file name.')
    g1 = p.add_argument_group(title='Relate to Hi-C data:')
    g1.add_argument('-p', '--path', help='Cooler URI (or a band archive .npz).')
    g1.add_argument('-C', '--chroms', nargs='*', default=['#', 'X'],
                    help='List of chromosome labels. "#" stands for chromosomes with numerical labels. '
                    '"--chroms" with zero argument will include all chromosome data.')
    g2 = p.add_argument_group(title='Algorithm Parameters:')
    g2.add_argument('--pw', type=int, nargs='+', help='List of the peak widths.')
    g2.add_argument('--ww', type=int, nargs='+', help='List of the donut widths.')
    g2.add_argument('--maxww', type=int, default=10, help='Maximum donut width.')
    g2.add_argument('--siglevel', type=float, default=0.05, help='Significant Level.')
    g2.add_argument('--sumq', type=float, default=0.01, help='Maximum sum of the 2 q-values of an isolated peak pixel.')
    g2.add_argument('--double-fold', type=float, default=1.75, help='Minimum fold enrichment over both backgrounds.')
    g2.add_argument('--single-fold', type=float, default=2, help='Minimum fold enrichment over either background.')
    g2.add_argument('--clr-weight-name', default='weight', help='Name of the weight column.')
    g2.add_argument('--use-raw', action='store_true', help='Sort peak pixels by raw signals during local clustering.')
    g2.add_argument('--min-marginal-peaks', type=int, default=2, help='Minimum marginal number of peaks of an anchor.')
    g2.add_argument('--min-local-reads', type=int, default=16, help='Minimum sum of contacts in the vicinity of a loop (at most 2097151 // (2 maxww + 1)**2: 4755 at --maxww 10).')
    g2.add_argument('--only-anchors', action='store_true', help='Either of the peak loci must be an anchor.')
    g2.add_argument('--maxapart', type=int, default=10000000, help='Maximum genomic distance between two loci.')
    g2.add_argument('--nproc', type=int, default=1, help='Number of worker processes (one per GPU; clamped to the GPUs present).')
    g2.add_argument('--device', type=int, default=None, help='GPU ordinal (default: local rank / worker index).')
    _add_deterministic(g2)
    return p


def _bhfdr_parser():
    """Flags of scripts/pyBHFDR:12-58."""
    from . import __version__, __reference__
    p = argparse.ArgumentParser(usage='%(prog)s <-O output> [options]',
                                description='BH-FDR peak calling on AMD MI355X (drop-in for pyBHFDR of %s).' % __reference__,
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-v', '--version', action='version', version=' '.join(['%(prog)s', __version__]))
    p.add_argument('-O', '--output', help='Output file name.')
    p.add_argument('--logFile', default='pyBHFDR.log', help='Logging file name.')
    g1 = p.add_argument_group(title='Relate to Hi-C data:')
    g1.add_argument('-p', '--path', help='Cooler URI (or a band archive .npz).')
    g1.add_argument('-C', '--chroms', nargs='*', default=['#', 'X'], help='List of chromosome labels.')
    g2 = p.add_argument_group(title='Algorithm Parameters:')
    g2.add_argument('--pw', type=int, default=2, help='Width of the interaction region surrounding the peak.')
    g2.add_argument('--ww', type=int, default=5, help='Width of the donut sampled.')
    g2.add_argument('--maxww', type=int, default=10, help='Maximum donut width.')
    g2.add_argument('--siglevel', type=float, default=0.05, help='Significant Level.')
    g2.add_argument('--maxapart', type=int, default=2000000, help='Maximum genomic distance between two loci.')
    g2.add_argument('--clr-weight-name', default='weight', help='Name of the weight column.')
    g2.add_argument('--nproc', type=int, default=1, help='Number of worker processes (one per GPU; clamped to the GPUs present).')
    g2.add_argument('--device', type=int, default=None, help='GPU ordinal.')
    _add_deterministic(g2)
    return p


def _setup_logging(logfile, rotating):
    """Root logger as in scripts/pyHICCUPS:88-105 / scripts/pyBHFDR:66-84."""
    logger = logging.getLogger()
    logger.setLevel(10)
    console = logging.StreamHandler()
    fh = (logging.handlers.RotatingFileHandler(logfile, maxBytes=200000, backupCount=5) if rotating
          else logging.FileHandler(logfile))
    console.setLevel('INFO')
    fh.setLevel('INFO')
    fmt = logging.Formatter(fmt='%(name)-21s %(levelname)-7s @ %(asctime)s: %(message)s', datefmt='%m/%d/%y %H:%M:%S')
    console.setFormatter(fmt)
    fh.setFormatter(fmt)
    logger.addHandler(console)
    logger.addHandler(fh)
    return logger


def select_chroms(chromnames, chroms):
    """scripts/pyHICCUPS:185-187."""
    out = []
    for key in chromnames:
        label = key.lstrip('chr')
        if (not chroms) or (label.isdigit() and '#' in chroms) or (label in chroms):
            out.append(key)
    return out


# HPK_CLI_TIMELINE=<file>: every stage of every chromosome as "thread stage chromosome start end" (seconds since the first
# event) - where a run's wall time goes (scripts/measure/gpu_r05_e2e_timeline.sh)
_TL = [] if os.environ.get('HPK_CLI_TIMELINE') else None


class _Stage:
    def __init__(self, what, key):
        self.what, self.key = what, key

    def __enter__(self):
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if _TL is not None:
            import threading
            _TL.append((threading.current_thread().name, self.what, str(self.key), self.t0, time.perf_counter()))


def _dump_timeline():
    if _TL:
        t0 = min(e[3] for e in _TL)
        with open(os.environ['HPK_CLI_TIMELINE'], 'w') as f:
            for th, what, key, a, b in sorted(_TL, key=lambda e: e[3]):
                f.write('%-12s %-8s %-6s %8.3f %8.3f\n' % (th, what, key, a - t0, b - t0))


def _read(args_dict, src, key, use_pixels):
    """What one chromosome needs from the file (the first half of worker(), scripts/pyHICCUPS:139-166, without the
    per-diagonal extraction) - HDF5 reads and chunk inflation only, no GPU call: runs on the reader thread."""
    num = args_dict['maxapart'] // src.binsize + args_dict['maxww'] + 1
    if use_pixels:
        return ('pixels', num) + tuple(src.fetch_pixels(key, args_dict['clr_weight_name']))
    return ('band', num) + tuple(src.fetch(key, num, args_dict['clr_weight_name']))


def _to_item(key, got, ctx):
    """-> (label, raw f32 [n, num] or a DeviceBand, weight f64 [n] or None, biases or None).  A sparse pixel table goes to
    the GPU as it is and the band is built there (hpk_devband_create: a fifth to a twentieth of the dense band's bytes over
    the bus); a band denser than one stored pixel in five cells is smaller than its pixel table: built on the host
    (hpk_band_from_coo, threaded) and uploaded with the batch."""
    label = key.lstrip('chr')
    if got[0] == 'band':
        _, num, raw, w, b = got
        return label, raw, w, b
    _, num, i, j, cnt, n, w, b = got
    if 20 * i.size <= 4 * n * num:
        return label, ctx.devband(i, j, cnt, n, num, w, b), None, None
    from . import band as _band
    return label, _band.band_from_coo(i, j, cnt, n, num), w, b


def _fetch(args_dict, src, key, ctx=None):
    """One chromosome's band as the library wants it, read and prepared in the calling thread."""
    use_pixels = ctx is not None and hasattr(src, 'fetch_pixels') and not os.environ.get('HPK_HOST_BANDS')
    return _to_item(key, _read(args_dict, src, key, use_pixels), ctx)


def _submit_group(args_dict, mode, items, device, res):
    """A group of fetched chromosomes onto the GPU's queue as one batch (hpk_submit_batch); returns a PendingBatch."""
    from . import callers, _lib
    a = args_dict
    ctx = _lib.default_context(device)
    if mode == 'hiccups':
        return callers.hiccups_batch_submit(items, pw=a['pw'], ww=a['ww'], maxww=a['maxww'], sig=a['siglevel'], sumq=a['sumq'],
                                            double_fold=a['double_fold'], single_fold=a['single_fold'], maxapart=a['maxapart'],
                                            res=res, use_raw=a['use_raw'], min_marginal_peaks=a['min_marginal_peaks'],
                                            onlyanchor=a['only_anchors'], min_local_reads=a['min_local_reads'], ctx=ctx)
    return callers.bhfdr_batch_submit(items, pw=a['pw'], ww=a['ww'], sig=a['siglevel'], maxww=a['maxww'],
                                      maxapart=a['maxapart'], res=res, ctx=ctx)


GROUP_BYTES = 3 << 29        # band bytes of one batch (1.5 GiB; the device workspaces of a batch are ~8x its bands)
GROUP_CHROMS = 8             # ... and chromosomes per batch


def _score_queue(args_dict, mode, queue, device, shared=False):
    """One GPU worker: takes chromosomes from the shared largest-first queue, hands them to the GPU in batches (several
    chromosomes per set of launches: up to GROUP_CHROMS chromosomes / GROUP_BYTES of bands, the first one alone so that the GPU
    starts at once), `pipeline_depth` batches in flight - while batch i is on the GPU, batch i + 1 is read and uploaded and
    batch i - 1 goes through clustering on the host; a batch is collected before the one after next is submitted, so the
    bounds a chromosome inherits (record bound, halo, survivors' bound: DESIGN 4.6-4.9) come from the batch before last
    at the latest.  -> {label: table}"""
    from . import io, _lib
    import collections
    import queue as _queue
    import threading
    pending, out = collections.deque(), {}

    # The reader: a thread of its own takes chromosomes from the queue and reads them (HDF5 + chunk inflation release the
    # GIL), two ahead at most, so that chromosome i + 1 is read while chromosome i's batch is on the GPU and chromosome
    # i - 1 goes through clustering.  It opens its own handle on the file and makes no GPU call (a context belongs to one thread).
    # (several workers on one queue - `shared` -: one chromosome ahead only, so that a worker does not sit on chromosomes
    # another GPU could start on; the largest-first order is what balances the tail)
    fetched = _queue.Queue(maxsize=1 if shared else 2)
    info = {}
    stop = threading.Event()            # the consumer gave up (an exception on its side): the reader takes no more chromosomes

    def put(item):
        while not stop.is_set():
            try:
                fetched.put(item, timeout=0.2)
                return True
            except _queue.Full:
                pass
        return False

    def reader():
        src = None
        try:
            src = io.open_source(args_dict['path'])
            info['binsize'] = src.binsize
            use_pixels = hasattr(src, 'fetch_pixels') and not os.environ.get('HPK_HOST_BANDS')
            if use_pixels and hasattr(src, 'enable_pool') and not os.environ.get('HPK_READ_NO_POOL'):
                info['release'] = src.enable_pool()       # (the consumer hands a chromosome's pixel columns back: _consume)
            for key in queue:
                if stop.is_set():
                    return
                with _Stage('read', key):
                    got = _read(args_dict, src, key, use_pixels)
                if not put((key, got)):
                    return
            put(None)
        except BaseException as e:          # (re-raised by the consumer)
            put(e)
        finally:
            close = getattr(src, 'close', None)
            if close is not None:
                close()

    th = threading.Thread(target=reader, name='hpk-reader', daemon=True)
    th.start()
    # The context - the HIP runtime's start-up, ~0.3 s in a fresh process - is created while the reader opens the file and reads the
    # first chromosome (in this thread: a context belongs to the thread that scores on it).
    ctx = None

    def _restore_mode():
        if ctx is not None and 'HPK_SPEC_HALO' not in os.environ:
            ctx.set_option('spec_halo', 2)          # _lib.default_context's own mode
    try:
        with _Stage('context', '-'):
            ctx = _lib.default_context(device)
        # every run starts without memory of the chromosomes an earlier run in this process scored
        ctx.set_option('reset_hints', 1)
        # The run's mode, on the process-wide context the drop-in hiccups() / bhfdr() score on as well: set for this run and put back
        # when it ends (_restore_mode), so that an in-process --history-dependent run does not leave those functions' values
        # depending on the calls before.  HPK_SPEC_HALO in the environment wins over either flag (it is what the library started with).
        if 'HPK_SPEC_HALO' not in os.environ:
            ctx.set_option('spec_halo', 1 if (args_dict.get('history_dependent') and not args_dict.get('deterministic')) else 2)
        depth = ctx.pipeline_depth
        return _consume(args_dict, mode, device, ctx, depth, fetched, info, pending, out, collections)
    finally:
        # whatever ended the loop: the reader stops taking chromosomes off the (shared) queue, is not left blocked on a full
        # hand-over queue, and closes its file
        stop.set()
        while th.is_alive():
            try:
                fetched.get_nowait()
            except _queue.Empty:
                pass
            th.join(timeout=0.05)
        _restore_mode()


def _consume(args_dict, mode, device, ctx, depth, fetched, info, pending, out, collections):
    from . import _lib

    # A batch is collected here (the wait for its kernels and the library's host half: C, this thread's context); the Python half of
    # its chromosomes - gap filter, combination, clustering - runs on a thread of its own, beside the next chromosomes' band
    # building and submission, which are C calls too (profiles/r05_cli_timeline.txt: it was half of this thread's time).
    from concurrent.futures import ThreadPoolExecutor
    finpool = ThreadPoolExecutor(1, thread_name_prefix='hpk-finish')
    finishing = []

    def finish(labels, half):
        with _Stage('finish', labels[0]):
            return list(zip(labels, half()))

    def collect():
        labels, call = pending.popleft()
        with _Stage('collect', labels[0]):
            half = call.collect()
        if os.environ.get('HPK_CLI_FINISH_INLINE'):      # (profiling: everything on this thread)
            for label, table in finish(labels, half):
                out[label] = table
            return
        finishing.append(finpool.submit(finish, labels, half))

    try:
        _consume_loop(args_dict, mode, device, ctx, depth, fetched, info, pending, collect, _lib)
        for fut in finishing:                       # (a finisher's exception surfaces here, like the reference's worker's would)
            for label, table in fut.result():
                out[label] = table
    finally:
        finpool.shutdown(wait=True, cancel_futures=True)
    _dump_timeline()
    return out


def _consume_loop(args_dict, mode, device, ctx, depth, fetched, info, pending, collect, _lib):
    group, nbytes = [], 0
    while True:
        # (nothing read yet: the host half of the oldest batch in flight - clustering, the tables - goes here instead of behind the
        # last chromosome; the bounds the next batches inherit only get fresher)
        while pending and fetched.empty():
            collect()
        with _Stage('wait', '-'):
            got = fetched.get()
        if isinstance(got, BaseException):
            raise got
        if got is not None:
            with _Stage('band', got[0]):
                item = _to_item(got[0], got[1], ctx)
                if got[1][0] == 'pixels' and info.get('release'):
                    info['release'](*got[1][2:5])           # the band is built (on the GPU or the host): the columns serve the next chromosome
            got = (got[0], None)
            group.append(item)
            nbytes += item[1].nbytes
        if group and (got is None or nbytes >= GROUP_BYTES or len(group) >= min(GROUP_CHROMS, _lib.HPK_MAX_BATCH) or not pending):
            if len(pending) >= depth:
                collect()
            with _Stage('submit', group[0][0]):
                pending.append(([g[0] for g in group], _submit_group(args_dict, mode, group, device, info['binsize'])))
            group, nbytes = [], 0
        if got is None:
            break
    while pending:
        collect()


def _gpu_count():
    """Visible GPUs (no torch needed: the ROCm runtime answers through libhpk's own dependency)."""
    import ctypes
    try:
        hip = ctypes.CDLL('libamdhip64.so')
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def worker_devices(nproc, device, ngpus):
    """The reference's --nproc counts CPU processes (scripts/pyHICCUPS:192-198); here a worker needs a GPU.  With
    --device every chromosome goes to that one GPU (one worker: a second process on the same GPU only time-slices);
    otherwise min(nproc, GPUs) workers, worker w on GPU w.  -> (workers, [device per worker])
    (HPK_CLI_SHARE_GPU=1 - tests on a one-GPU box - keeps --nproc workers and lets them share the GPUs there are.)"""
    if os.environ.get('HPK_CLI_SHARE_GPU') and nproc > 1:
        g = max(1, ngpus)
        return nproc, [device if device is not None else w % g for w in range(nproc)]
    if device is not None:
        return 1, [device]
    n = max(1, min(nproc, ngpus)) if ngpus > 0 else nproc      # no GPU visible: let hpk_create report it
    return n, list(range(n))


def _gpu_worker(args_dict, mode, device, sizes, value, results, wid=0):
    """One worker process = one GPU: drains the shared queue (a counter over the same largest-first list in every
    process) and sends back its {label: table}, or what stopped it as (type name, status code, message) strings - an
    exception object may fail to pickle in the queue's feeder thread, and the parent would wait for ever."""
    from . import parallel
    try:
        queue = parallel.WorkQueue(sizes, parallel.mp_counter(value))
        results.put((wid, _score_queue(args_dict, mode, queue, device, shared=True), None))
    except BaseException as e:
        results.put((wid, None, _describe_error(e)))


def _describe_error(e):
    """An exception as plain strings / integers (always picklable): (type name, library status or None, message)"""
    return (type(e).__name__, getattr(e, 'status', None), getattr(e, 'msg', None) or str(e))


def _rebuild_error(desc):
    """(type name, library status, message) of a worker's failure -> the exception the single-process path would have raised"""
    from . import _lib
    name, status, msg = desc
    if name == 'EmptyStepError':
        return _lib.EmptyStepError(status if status is not None else _lib.ERR_EMPTY_STEP, msg)
    if name == 'HpkError':
        return _lib.HpkError(status if status is not None else -1, msg)
    return RuntimeError('GPU worker failed: %s: %s' % (name, msg))


def run_workers(args_dict, mode, sizes, devices, poll=1.0):
    """--nproc N: one process per GPU around one queue (counterpart of Pool(nproc).map(worker, Params),
    scripts/pyHICCUPS:192-198) -> {label: table} of all chromosomes.  A worker that dies without reporting (a crash in the
    native library, the kernel's out-of-memory killer) is noticed by its exit code; the others are stopped."""
    import multiprocessing as mp
    import queue as _queue
    ctxmp = mp.get_context('spawn')
    value = ctxmp.Value('i', 0)
    results = ctxmp.Queue()
    procs = [ctxmp.Process(target=_gpu_worker, args=(args_dict, mode, d, sizes, value, results, w)) for w, d in enumerate(devices)]
    for p in procs:
        p.start()
    tables, err, reported = {}, None, set()
    try:
        while len(reported) < len(procs):
            try:
                wid, part, e = results.get(timeout=poll)
            except _queue.Empty:
                dead = [w for w, p in enumerate(procs) if w not in reported and not p.is_alive()]
                if dead:
                    try:                                    # its result may have arrived between the time-out and the check
                        wid, part, e = results.get(timeout=poll)
                    except _queue.Empty:
                        raise RuntimeError('GPU worker %d (device %s) exited with code %s without reporting a result' % (
                            dead[0], devices[dead[0]], procs[dead[0]].exitcode))
                else:
                    continue
            reported.add(wid)
            if e is not None:
                err = err or _rebuild_error(e)
            else:
                tables.update(part)
    finally:
        for p in procs:
            if len(reported) < len(procs) and p.is_alive():
                p.terminate()
            p.join()
    if err is not None:
        raise err
    return tables


def _run(mode, argv):
    from . import io, parallel
    parser = _hiccups_parser() if mode == 'hiccups' else _bhfdr_parser()
    commands = list(sys.argv[1:] if argv is None else argv)
    if not commands:
        commands.append('-h')                       # scripts/pyHICCUPS:76-78
    args = parser.parse_args(commands)
    logger = _setup_logging(args.logFile, rotating=(mode == 'bhfdr'))
    logger.info('Python Version: {}'.format(sys.version.split()[0]))
    logger.info('\n# ARGUMENT LIST:\n' + '\n'.join('# {0} = {1}'.format(k, v) for k, v in sorted(vars(args).items())))
    if mode == 'hiccups' and (not args.pw or not args.ww):
        parser.error('--pw and --ww are required')
    logger.info('Loading Hi-C data ...')
    with _Stage('open', '-'):
        src = io.open_source(args.path)
        res = src.binsize
        keys = select_chroms(src.chromnames, args.chroms)
        sizes = {k: src.nbins(k) for k in keys}
    a = vars(args)
    rank, world, local = parallel.dist_env()
    logger.info('Calling Peaks ...')
    logger.info('Tile geometry: {0}'.format('inherited from the chromosomes scored before, no second pass (E / p / q can differ in the 14th '
                                            'digit between runs that order or batch the chromosomes differently)'
                                            if args.history_dependent and not args.deterministic else
                                            'every chromosome under the layout of its own frozen width (results independent of chromosome '
                                            'order, batching and the number of workers; --history-dependent skips the second pass)'))
    if world > 1:                                    # torchrun: one rank per GPU, one queue, tables gathered on rank 0
        import torch.distributed as dist
        dist.init_process_group('gloo')              # only the queue's counter and Python objects travel
        dev = local if args.device is None else args.device
        queue = parallel.WorkQueue(sizes, parallel.store_counter())
        try:                                         # a rank that fails still takes part in the gather: its error travels instead of its tables
            mine = _score_queue(a, mode, queue, dev, shared=world > 1)
        except Exception as e:
            mine = {'__error__': _describe_error(e)}
        tables = parallel.gather_tables(mine, rank, world)
        dist.destroy_process_group()
        if rank != 0:
            if '__error__' in mine:
                raise _rebuild_error(mine['__error__'])
            return 0
        if '__error__' in tables:
            raise _rebuild_error(tables['__error__'])
    else:
        # Pool.map over GPU workers (scripts/pyHICCUPS:192-198); one worker runs in this process
        nworkers, devices = worker_devices(args.nproc, args.device, _gpu_count() if args.nproc > 1 else 0)
        if args.nproc > 1:
            logger.info('--nproc {0}: {1} worker(s) on GPU(s) {2}'.format(args.nproc, nworkers, devices))
        if nworkers > 1:
            tables = run_workers(a, mode, sizes, devices)
        else:
            queue = parallel.WorkQueue(sizes, parallel.local_counter())
            tables = _score_queue(a, mode, queue, devices[0] if args.device is not None else 0)
    with open(args.output, 'w') as out:
        for key in keys:                             # the order of the chromosomes in the file, as the reference writes them
            label = key.lstrip('chr')
            table = tables[label]
            out.write(format_hiccups(label, table, res) if mode == 'hiccups' else format_bhfdr(label, table, res))
    logger.info('Done!')
    return 0


def main_hiccups(argv=None):
    return _run('hiccups', argv)


def main_bhfdr(argv=None):
    return _run('bhfdr', argv)
