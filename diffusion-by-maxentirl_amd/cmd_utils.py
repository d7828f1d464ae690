"""`--a.b.c value` command-line overrides -> nested dict (same behaviour as the reference's
cmd_utils.py:16-61; part of the train_*/generate_* CLI surface)."""
import ast


def parse_arg_type(val):
    """'3' -> 3, '1e-6' -> 1e-06, 'true' -> True, 'null'/'none' -> None, '[1,2]' -> [1, 2], else str."""
    if val.isnumeric():
        return int(val)
    try:
        return float(val)
    except ValueError:
        pass
    low = val.lower()
    if low in ("true", "false"):
        return low == "true"
    if low in ("null", "none"):
        return None
    if val.startswith("[") and val.endswith("]"):
        return ast.literal_eval(val)  # the reference eval()s; literal_eval accepts the same list syntax safely
    return val


def parse_unknown_args(l_args):
    """['--training.lr', '1e-6', ...] -> {'training.lr': 1e-06, ...}; '=' forms are rejected."""
    out = {}
    for i in range(len(l_args) // 2):
        key, val = l_args[2 * i], l_args[2 * i + 1]
        assert "=" not in key, "optional arguments should be separated by space"
        out[key.strip("-")] = parse_arg_type(val)
    return out


def parse_nested_args(d_cmd_cfg):
    """{'a.b.c': 1} -> {'a': {'b': {'c': 1}}}."""
    out = {}
    for key, val in d_cmd_cfg.items():
        parts = key.split(".")
        d = out
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = val
    return out
