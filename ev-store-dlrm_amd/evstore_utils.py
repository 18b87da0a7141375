"""Mirror of the reference's evstore_utils.py (the file formats the inference forks read and write around the hot path):

  load_new_ev_table(ld_model, ev_path)            :13-29   ev-table-{1..26}.csv (header line, then one row per line) -> the
                                                           model's emb_l.N.weight tensors; here also straight into HBM:
                                                           ev_tables_from_csv_dir(ev_path) -> EVTables
  store_training_config / read_training_config    :31-52   training_config.txt, 6 text lines (ln_emb, m_den, batch counts)
  prepare_inference_trace_folder                  :54-58
  write_inf_workload_to_file                      :60-73   workload-group-{1..26}.csv: header G{k}_key, then one key per
                                                           request (keys as apply_emb_evstore forms them: "table-row")
  read_inf_workload(dir)                                   (not in the reference: the traces back as a (n, 26) int32 array of
                                                           row ids -- what ev_lookup / the batched lookups take -- to replay
                                                           a recorded workload through the cache tiers)

Same names, arguments, prints and bytes as the reference (tests/golden/evstore_utils.npz holds what ITS functions wrote).
Host-side plumbing only."""
import ast
import os
from pathlib import Path

import numpy as np
import torch

TRAINING_CONFIG_FILE = "training_config.txt"


def load_new_ev_table(ld_model, ev_path):
    """evstore_utils.py:13-29: replace the 26 embedding tensors of a loaded checkpoint by the CSV tables under ev_path"""
    import pandas as pd
    print("Load new set of EV Table from = " + ev_path)
    for ev_idx in range(0, 26):
        new_ev_path = os.path.join(ev_path, "ev-table-" + str(ev_idx + 1) + ".csv")
        new_ev_arr = pd.read_csv(new_ev_path, dtype=float, delimiter=',').to_numpy()
        new_ev_tensor = torch.FloatTensor(new_ev_arr)
        print("Loading NEW EV per embedding layer = " + new_ev_path)
        ld_model["state_dict"][str("emb_l." + str(ev_idx) + ".weight")] = new_ev_tensor
    print("Done loading all EV-Table from " + ev_path)


def ev_tables_from_csv_dir(ev_path, n_tables=26, device="cuda"):
    """the same CSV tables (what load_new_ev_table reads) straight into HBM as this package's EVTables"""
    import pandas as pd
    from .dlrm_ops import EVTables
    ws = [torch.FloatTensor(pd.read_csv(os.path.join(ev_path, "ev-table-%d.csv" % (k + 1)), dtype=float, delimiter=',').to_numpy())
          for k in range(n_tables)]
    return EVTables.from_fp32(ws, device=device)


def store_training_config(file_path, table_feature_map, nbatches, nbatches_test, ln_emb, m_den):
    """evstore_utils.py:31-40"""
    with open(file_path, 'w') as f:
        f.write('The order of the arguments: table_feature_map, nbatches, nbatches_test, ln_emb, m_den\n')
        f.write(str(table_feature_map) + "\n")
        f.write(str(nbatches) + "\n")
        f.write(str(nbatches_test) + "\n")
        f.write(str(ln_emb.tolist()) + "\n")
        f.write(str(m_den) + "\n")
    print("Done writing training config to : " + file_path + "\n")


def read_training_config(file_path):
    """evstore_utils.py:42-52 -> table_feature_map, nbatches, nbatches_test, ln_emb (np.array), m_den"""
    print("Read training config from : " + file_path)
    with open(file_path) as f:
        lines = [line.rstrip() for line in f]
    table_feature_map = ast.literal_eval(lines[1])
    nbatches = int(lines[2])
    nbatches_test = int(lines[3])
    ln_emb = np.array(ast.literal_eval(lines[4]))
    m_den = int(lines[5])
    return table_feature_map, nbatches, nbatches_test, ln_emb, m_den


def prepare_inference_trace_folder(input_data_name, percent_data_for_inference):
    """evstore_utils.py:54-58"""
    print("Create folder to store the model and ev-tables")
    outdir = os.path.join("logs", "inf-workload-traces", input_data_name, "inference=" + str(percent_data_for_inference))
    Path(outdir).mkdir(parents=True, exist_ok=True)
    return outdir


def write_inf_workload_to_file(workload_traces_outdir, arr_inference_workload):
    """evstore_utils.py:60-73, byte for byte: one file per table, header G{k}_key, the k-th key of every request"""
    print("Total inference = " + str(len(arr_inference_workload)))
    arrfile = []
    for idx in range(0, 26):
        arrfile.append(open(workload_traces_outdir + "/workload-group-" + str(idx + 1) + ".csv", 'w'))
        arrfile[idx].write("G" + str(idx + 1) + "_key\n")
    for grouped_keys in arr_inference_workload:
        id = 0
        for key in grouped_keys:
            arrfile[id].write(key + "\n")
            id += 1
    for f in arrfile:   # (the reference leaves them to the garbage collector: the same bytes on disk)
        f.close()


def read_inf_workload(workload_traces_dir, n_tables=26):
    """the traces write_inf_workload_to_file wrote -> (n_requests, n_tables) int32 row ids: the request stream of ev_lookup /
    evs_cache_request / the batched lookups.  Keys are "table-row" (table 1-based); they are sorted into their table's column
    by that prefix, in file order."""
    cols = [[] for _ in range(n_tables)]
    for k in range(n_tables):
        with open(os.path.join(workload_traces_dir, "workload-group-%d.csv" % (k + 1))) as f:
            lines = [ln.strip() for ln in f if ln.strip()]
        assert lines and lines[0] == "G%d_key" % (k + 1), "workload-group-%d.csv: unexpected header %r" % (k + 1, lines[:1])
        for key in lines[1:]:
            t, _, r = key.partition("-")
            assert 1 <= int(t) <= n_tables, "workload-group-%d.csv: key %r" % (k + 1, key)
            cols[int(t) - 1].append(int(r))
    n = len(cols[0])
    assert all(len(c) == n for c in cols), "the traces hold different numbers of keys per table"
    return np.asarray(cols, dtype=np.int32).T.copy() if n else np.zeros((0, n_tables), np.int32)
