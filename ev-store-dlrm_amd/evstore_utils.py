"""The side files of the inference forks, written from their FORMATS (reference: evstore_utils.py:13-73 defines them; SURVEY
§8 asks for read_training_config's format, the rest is what the tiers need to be fed a recorded model / workload):

  training_config.txt       six text lines: a free-text header naming the fields, then repr(table_feature_map) (a dict),
                            nbatches, nbatches_test, the ln_emb list, m_den -- one per line (evstore_utils.py:31-52)
  ev-table-{1..T}.csv       one header line, then one comma-separated row per line (:13-29)
  workload-group-{1..T}.csv header "G{k}_key", then the k-th key of every request, one per line; keys are "table-row" with
                            the table 1-based (:60-73, keys as dlrm_s_pytorch_C1.py:236-240 forms them)

The functions keep the reference's names and argument order so the forks' call sites read the same; the bodies are a small
field table and column-wise writers, not the reference's statements.  Host-side plumbing only."""
import ast
import os

import numpy as np
import torch

TRAINING_CONFIG_FILE = "training_config.txt"
N_TABLES = 26

# the config file as a record: (field, text -> value, value -> text), in file order behind the header line
_CONFIG_HEADER = "The order of the arguments: table_feature_map, nbatches, nbatches_test, ln_emb, m_den"
_CONFIG_FIELDS = (
    ("table_feature_map", ast.literal_eval, str),
    ("nbatches", int, str),
    ("nbatches_test", int, str),
    ("ln_emb", lambda s: np.array(ast.literal_eval(s)), lambda v: str(np.asarray(v).tolist())),
    ("m_den", int, str),
)


def format_training_config(**fields):
    """the six lines of training_config.txt as one string"""
    return "".join(line + "\n" for line in [_CONFIG_HEADER] + [show(fields[name]) for name, _, show in _CONFIG_FIELDS])


def parse_training_config(text):
    """the text of a training_config.txt -> (table_feature_map, nbatches, nbatches_test, ln_emb as np.ndarray, m_den)"""
    body = text.splitlines()[1:1 + len(_CONFIG_FIELDS)]
    if len(body) != len(_CONFIG_FIELDS):
        raise ValueError("training config: %d value lines behind the header, expected %d" % (len(body), len(_CONFIG_FIELDS)))
    return tuple(read(line.rstrip()) for (_, read, _), line in zip(_CONFIG_FIELDS, body))


def store_training_config(file_path, table_feature_map, nbatches, nbatches_test, ln_emb, m_den):
    with open(file_path, "w") as f:
        f.write(format_training_config(table_feature_map=table_feature_map, nbatches=nbatches, nbatches_test=nbatches_test,
                                       ln_emb=ln_emb, m_den=m_den))


def read_training_config(file_path):
    with open(file_path) as f:
        return parse_training_config(f.read())


def _table_csv(ev_path, k):
    return os.path.join(ev_path, "ev-table-%d.csv" % (k + 1))


def _read_table_csv(path):
    """one ev-table CSV (header line skipped) -> (rows, d) fp32 tensor; values parsed as doubles, then rounded to fp32 once
    (what the reference's pd.read_csv(dtype=float) + torch.FloatTensor does).  pandas' C reader when pandas is importable --
    the 10 M-row Kaggle tables are gigabytes of text --, numpy's loadtxt otherwise: the same doubles either way."""
    try:
        import pandas as pd
    except ImportError:
        pd = None
    if pd is not None:
        arr = pd.read_csv(path, dtype=np.float64, delimiter=",", header=0).to_numpy(dtype=np.float64)
    else:
        arr = np.loadtxt(path, dtype=np.float64, delimiter=",", skiprows=1, ndmin=2)
    return torch.from_numpy(np.ascontiguousarray(arr.astype(np.float32)))


def load_new_ev_table(ld_model, ev_path, n_tables=N_TABLES):
    """the CSV tables under ev_path become the emb_l.{k}.weight entries of a loaded checkpoint's state_dict"""
    ld_model["state_dict"].update(("emb_l.%d.weight" % k, _read_table_csv(_table_csv(ev_path, k))) for k in range(n_tables))


def ev_tables_from_csv_dir(ev_path, n_tables=N_TABLES, device="cuda"):
    """the same CSV tables straight into HBM as this package's EVTables"""
    from .dlrm_ops import EVTables
    return EVTables.from_fp32([_read_table_csv(_table_csv(ev_path, k)) for k in range(n_tables)], device=device)


def _trace_csv(outdir, k):
    return os.path.join(outdir, "workload-group-%d.csv" % (k + 1))


def prepare_inference_trace_folder(input_data_name, percent_data_for_inference):
    outdir = os.path.join("logs", "inf-workload-traces", input_data_name, "inference=%s" % (percent_data_for_inference,))
    os.makedirs(outdir, exist_ok=True)
    return outdir


def write_inf_workload_to_file(workload_traces_outdir, arr_inference_workload, n_tables=N_TABLES):
    """requests (each a sequence of at most n_tables keys) -> the n_tables trace files, written a column at a time.  Key j of a
    request goes to file j, as the reference writes it (evstore_utils.py:68-73): a request with FEWER keys leaves the later
    files one line shorter, a request with MORE keys than files is an IndexError -- nothing is dropped silently."""
    columns = [[] for _ in range(n_tables)]
    for i, keys in enumerate(arr_inference_workload):
        keys = list(keys)
        if len(keys) > n_tables:
            raise IndexError("write_inf_workload_to_file: request %d holds %d keys, there are %d trace files" % (i, len(keys), n_tables))
        for k, key in enumerate(keys):
            columns[k].append(key)
    for k, keys in enumerate(columns):
        with open(_trace_csv(workload_traces_outdir, k), "w") as f:
            f.write("".join(s + "\n" for s in ["G%d_key" % (k + 1)] + keys))


def read_inf_workload(workload_traces_dir, n_tables=N_TABLES):
    """the trace files -> (n_requests, n_tables) int32 row ids: the request stream of ev_lookup / evs_cache_request / the
    batched lookups.  A key goes to the column its "table-" prefix names, in file order."""
    cols = [[] for _ in range(n_tables)]
    for k in range(n_tables):
        with open(_trace_csv(workload_traces_dir, k)) as f:
            lines = [ln.strip() for ln in f if ln.strip()]
        if not lines or lines[0] != "G%d_key" % (k + 1):
            raise ValueError("workload-group-%d.csv: unexpected header %r" % (k + 1, lines[:1]))
        for key in lines[1:]:
            t, _, r = key.partition("-")
            if not 1 <= int(t) <= n_tables:
                raise ValueError("workload-group-%d.csv: key %r" % (k + 1, key))
            cols[int(t) - 1].append(int(r))
    n = len(cols[0])
    if any(len(c) != n for c in cols):
        raise ValueError("the traces hold different numbers of keys per table")
    return np.asarray(cols, dtype=np.int32).T.copy() if n else np.zeros((0, n_tables), np.int32)
