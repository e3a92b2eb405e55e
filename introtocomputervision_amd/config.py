"""Reader for the reference's run configurations (config/psN.yaml, read with yaml-cpp by each
problem set's Config class: ps4_cpp/lib/Config.cpp:25-133, ps1_cpp/src/Config.cpp,
ps2_cpp/lib/Config.cpp, ps5_cpp/lib/Config.cpp).  The files use a small YAML subset -- `---` / `...`
markers, comments, `key: scalar` pairs, maps nested by indentation (three deep in config/ps7.yaml) and
block sequences of scalars -- which is all this reader accepts (the C++ twin is shim/micv_config.hpp).  Scalars stay strings until asked
for with a type, exactly as YAML::Node::as<T>() works."""


class ConfigError(ValueError):
    pass


def _strip_comment(line):
    out, quote = [], None
    for i, ch in enumerate(line):
        if quote:
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
        elif ch == "#" and (i == 0 or line[i - 1] in " \t"):
            break
        out.append(ch)
    return "".join(out).rstrip()


def _scalar(text):
    text = text.strip()
    if len(text) >= 2 and text[0] == text[-1] and text[0] in "\"'":
        return text[1:-1]
    return text


class Node(dict):
    """A YAML map: values are strings (scalars), Nodes (nested maps, any depth) or lists of strings
    (block sequences of scalars, e.g. the per-trial frame counts of config/ps7.yaml:7-40)."""

    def has(self, key):
        return key in self

    def child(self, key):
        v = self.get(key)
        if not isinstance(v, Node):
            raise ConfigError(f"'{key}' is not a map")
        return v

    def seq(self, key):
        v = self.get(key)
        if not isinstance(v, list):
            raise ConfigError(f"'{key}' is not a sequence")
        return v

    def _raw(self, key):
        if key not in self:
            raise ConfigError(f"key '{key}' not found")
        v = self[key]
        if isinstance(v, Node):
            raise ConfigError(f"'{key}' is a map, not a scalar")
        if isinstance(v, list):
            raise ConfigError(f"'{key}' is a sequence, not a scalar")
        return v

    def as_str(self, key):
        return self._raw(key)

    def as_int(self, key):
        v = self._raw(key)
        try:
            return int(v, 0)
        except ValueError:
            raise ConfigError(f"'{key}: {v}' is not an integer") from None

    def as_float(self, key):
        v = self._raw(key)
        try:
            return float(v)
        except ValueError:
            raise ConfigError(f"'{key}: {v}' is not a number") from None

    def as_bool(self, key):
        v = self._raw(key).lower()
        if v in ("true", "yes", "on", "y"):
            return True
        if v in ("false", "no", "off", "n"):
            return False
        raise ConfigError(f"'{key}: {v}' is not a boolean")


def _is_item(body):
    return body == "-" or body.startswith("- ")


def loads(text):
    """Block maps nested to any depth by indentation and block sequences of scalars; a sequence may sit at
    its parent key's own indentation, as YAML allows.  Flow collections, anchors, multi-line scalars and
    sequences of maps are not part of the reference's files and are rejected."""
    lines = []  # (lineno, indent, body)
    for lineno, raw in enumerate(text.splitlines(), 1):
        line = _strip_comment(raw)
        if not line.strip() or line.strip() in ("---", "..."):
            continue
        if "\t" in line[:len(line) - len(line.lstrip())]:
            raise ConfigError(f"line {lineno}: tabs are not allowed for indentation")
        lines.append((lineno, len(line) - len(line.lstrip(" ")), line.strip()))
    pos = 0

    def parse_seq(indent):
        nonlocal pos
        out = []
        while pos < len(lines) and lines[pos][1] == indent and _is_item(lines[pos][2]):
            lineno, _, body = lines[pos]
            item = body[1:].strip()
            if item == "" or (":" in item and not (item[0] in "\"'")) and _split_key(item, lineno, probe=True):
                raise ConfigError(f"line {lineno}: only sequences of scalars are supported")
            out.append(_scalar(item))
            pos += 1
        if pos < len(lines) and lines[pos][1] > indent:
            raise ConfigError(f"line {lines[pos][0]}: unexpected indentation inside a sequence")
        return out

    def parse_map(indent):
        nonlocal pos
        node = Node()
        while pos < len(lines) and lines[pos][1] == indent:
            lineno, _, body = lines[pos]
            if _is_item(body):
                raise ConfigError(f"line {lineno}: sequence entry inside a map")
            key, value = _split_key(body, lineno)
            pos += 1
            if value != "":
                node[key] = _scalar(value)
                if pos < len(lines) and lines[pos][1] > indent:
                    raise ConfigError(f"line {lines[pos][0]}: indented entry below the scalar '{key}'")
                continue
            if pos < len(lines) and lines[pos][1] > indent:
                child_indent = lines[pos][1]
                node[key] = parse_seq(child_indent) if _is_item(lines[pos][2]) else parse_map(child_indent)
                if pos < len(lines) and lines[pos][1] > indent:
                    raise ConfigError(f"line {lines[pos][0]}: inconsistent indentation below '{key}'")
            elif pos < len(lines) and lines[pos][1] == indent and _is_item(lines[pos][2]):
                node[key] = parse_seq(indent)
            else:
                node[key] = Node()  # `key:` with nothing below it
        return node

    if lines and lines[0][1] != 0:
        raise ConfigError(f"line {lines[0][0]}: indented entry without a parent map")
    root = parse_map(0)
    if pos < len(lines):
        raise ConfigError(f"line {lines[pos][0]}: inconsistent indentation")
    return root


def _split_key(body, lineno, probe=False):
    if ":" not in body:
        if probe:
            return None
        raise ConfigError(f"line {lineno}: expected 'key: value'")
    key, _, value = body.partition(":")
    if value and not value.startswith((" ", "\t")):
        # a colon inside the key (e.g. a path) -- keys of the reference's files never contain one
        if probe:
            return None
        raise ConfigError(f"line {lineno}: expected a space after ':'")
    return _scalar(key), value.strip()


def dumps(node, prefix=""):
    """Every leaf as `path/to/key=value` (sequence entries `key[i]=value`), keys in byte order: the form
    tests compare across the two readers (Node::dump in shim/micv_config.hpp)."""
    out = []
    for k in sorted(node, key=lambda s: s.encode()):
        v = node[k]
        if isinstance(v, Node):
            if not v:
                out.append(f"{prefix}{k}={{}}\n")
            out.append(dumps(v, prefix + k + "/"))
        elif isinstance(v, list):
            out.extend(f"{prefix}{k}[{i}]={x}\n" for i, x in enumerate(v))
        else:
            out.append(f"{prefix}{k}={v}\n")
    return "".join(out)


def load(path):
    with open(path) as f:
        return loads(f.read())


def harris_params(cfg, section):
    """Config::Harris (ps4_cpp/lib/Config.cpp:43-54): the six keys of harris_trans / harris_sim."""
    n = cfg.child(section)
    return {
        "sobel_kernel_size": n.as_int("sobel_kernel_size"),
        "window_size": n.as_int("window_size"),
        "gaussian_sigma": n.as_float("gaussian_sigma"),
        "alpha": n.as_float("alpha"),
        "response_threshold": n.as_float("response_threshold"),
        "min_distance": n.as_int("min_distance"),
    }


def edge_params(cfg, section):
    """Config::EdgeDetect (ps1): gaussian_size, gaussian_sigma, lower/upper_threshold, sobel_aperture_size."""
    n = cfg.child(section)
    return {k: (n.as_float(k) if k == "gaussian_sigma" else n.as_int(k))
            for k in ("gaussian_size", "gaussian_sigma", "lower_threshold", "upper_threshold", "sobel_aperture_size")}


def hough_params(cfg, section):
    """Config::Hough (ps1): rho_bin_size, theta_bin_size, num_peaks, threshold."""
    n = cfg.child(section)
    return {k: n.as_int(k) for k in ("rho_bin_size", "theta_bin_size", "num_peaks", "threshold")}


def disparity_params(cfg, section):
    """Config::DisparitySSD / NCorr (ps2): window_radius, disparity_range."""
    n = cfg.child(section)
    return {k: n.as_int(k) for k in ("window_radius", "disparity_range")}


def mhi_params(cfg, section):
    """Config::MHI (ps7_cpp/lib/Config.cpp:35-47): diff_threshold, pre_blur_size (a square cv::Size),
    pre_blur_sigma, tau; `last_frame` is in the file (config/ps7.yaml:47) but the reference never reads it."""
    n = cfg.child(section)
    return {"diff_threshold": n.as_float("diff_threshold"), "pre_blur_size": n.as_int("pre_blur_size"),
            "pre_blur_sigma": n.as_float("pre_blur_sigma"), "tau": n.as_int("tau")}


def last_frames(cfg):
    """Config::loadActionLengths (ps7_cpp/lib/Config.cpp:49-66): a map of maps of maps of sequences ->
    {"PS7A<action>P<person>T<trial>": last frame}, actions / persons / trials numbered from 1."""
    actions = cfg.child("last_frame_of_action")
    out = {}
    for a in range(1, len(actions) + 1):
        persons = actions.child(f"action{a}")
        for p in range(1, len(persons) + 1):
            for t, frames in enumerate(persons.seq(f"person{p}"), 1):
                try:
                    out[f"PS7A{a}P{p}T{t}"] = int(frames, 0)
                except ValueError:
                    raise ConfigError(f"action{a}/person{p}: '{frames}' is not an integer") from None
    return out


def lk_params(cfg):
    """ps5's flat keys (ps5_cpp/lib/Config.cpp; config/ps5.yaml:11-17)."""
    return {k: cfg.as_int(k) for k in ("lk_window_size_1", "lk_window_size_3", "pyr_level_3-a", "pyr_level_3-b",
                                       "lk_window_size_4")}


def hough_circle_params(cfg, section):
    """Config::HoughCircle (ps1): min_radius, max_radius, num_peaks, threshold."""
    n = cfg.child(section)
    return {k: n.as_int(k) for k in ("min_radius", "max_radius", "num_peaks", "threshold")}
