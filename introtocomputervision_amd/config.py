"""Reader for the reference's run configurations (config/psN.yaml, read with yaml-cpp by each
problem set's Config class: ps4_cpp/lib/Config.cpp:25-133, ps1_cpp/src/Config.cpp,
ps2_cpp/lib/Config.cpp, ps5_cpp/lib/Config.cpp).  The files use a small YAML subset -- `---` / `...`
markers, comments, `key: scalar` pairs and one level of nested maps by two-space indentation -- which
is all this reader accepts (the C++ twin is shim/micv_config.hpp).  Scalars stay strings until asked
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
    """A YAML map: values are strings (scalars) or Nodes (nested maps)."""

    def has(self, key):
        return key in self

    def child(self, key):
        v = self.get(key)
        if not isinstance(v, Node):
            raise ConfigError(f"'{key}' is not a map")
        return v

    def _raw(self, key):
        if key not in self:
            raise ConfigError(f"key '{key}' not found")
        v = self[key]
        if isinstance(v, Node):
            raise ConfigError(f"'{key}' is a map, not a scalar")
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


def loads(text):
    root = Node()
    current, current_indent = None, 0
    for lineno, raw in enumerate(text.splitlines(), 1):
        line = _strip_comment(raw)
        if not line.strip() or line.strip() in ("---", "..."):
            continue
        if "\t" in line[:len(line) - len(line.lstrip())]:
            raise ConfigError(f"line {lineno}: tabs are not allowed for indentation")
        indent = len(line) - len(line.lstrip(" "))
        body = line.strip()
        if ":" not in body:
            raise ConfigError(f"line {lineno}: expected 'key: value'")
        key, _, value = body.partition(":")
        if value and not value.startswith((" ", "\t")):
            # a colon inside the key (e.g. a path) -- keys of the reference's files never contain one
            raise ConfigError(f"line {lineno}: expected a space after ':'")
        key = _scalar(key)
        if indent == 0:
            if value.strip() == "":
                current = Node()
                current_indent = None
                root[key] = current
            else:
                root[key] = _scalar(value)
                current = None
        else:
            if current is None:
                raise ConfigError(f"line {lineno}: indented entry without a parent map")
            if current_indent is None:
                current_indent = indent
            if indent != current_indent:
                raise ConfigError(f"line {lineno}: only one level of nesting is supported")
            if value.strip() == "":
                raise ConfigError(f"line {lineno}: nested maps below '{key}' are not supported")
            current[key] = _scalar(value)
    return root


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
