"""Longest dependent chain of a captured step: the DAG of a HIP graph dump (LSTEP_GRAPH_DOT=<file> with DEBUG_HIP_GRAPH_DOT_PRINT=1, see
tools/graph_dot_summary.py) weighted with the average run time of every kernel from a per-iteration kernel table (tools/prof_summary.py output)
plus a fixed cost per dependency edge.  Two answers:
  * the DAG's critical path (what an ideal executor with unlimited queues would be bound by);
  * the path when nodes the runtime put on the same replay stream also run in that stream's order (what the replay is bound by).
usage: python tools/graph_critical_path.py <dot file> <kernel_trace_per_iter.txt> [--edge-us 3.0]"""
import re
import sys


def demangle_key(name: str) -> str:
    """A loose key shared by the DOT's (mangled, truncated) names and the trace's demangled ones: the longest identifier-like piece."""
    name = re.sub(r"^_ZN5lstep\d+", "", name)
    name = re.sub(r"^_ZN2at6native\d+", "", name)
    name = re.sub(r"^rocprim\d+ROCPRIM_\d+_NS\d+detail\d+", "", name)      # rocprim17ROCPRIM_400200_NS6detail17trampoline_kernel...
    name = re.sub(r"^_Z[NL]?\d*", "", name)
    name = re.sub(r"^rocprim\d+ROCPRIM_\d+_NS\d+detail\d+", "", name)
    m = re.match(r"[A-Za-z_][A-Za-z_0-9]*", name)
    return (m.group(0) if m else name)[:40]


def main():
    dot, table = sys.argv[1], sys.argv[2]
    edge_us = float(sys.argv[sys.argv.index("--edge-us") + 1]) if "--edge-us" in sys.argv else 3.0
    txt = open(dot).read()
    nodes = {}
    for m in re.finditer(r'"(graph_\d+_node_(\d+))"\[[^\]]*?label="\d+\n([^\n]*)\nStreamId:(\d+)', txt):
        nodes[m.group(1)] = (int(m.group(2)), m.group(3), int(m.group(4)))
    edges = [(a, b) for a, b in re.findall(r'"(graph_\d+_node_\d+)" -> "(graph_\d+_node_\d+)"', txt) if a in nodes and b in nodes]
    # average run time per kernel name from the per-iteration table: "us/iter calls/iter avg us  kernel"
    tot = {}
    for line in open(table):
        m = re.match(r"\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        if not m or line.lstrip().startswith("#"):
            continue
        name = re.sub(r"^(void|static)\s+", "", m.group(4).strip())
        name = re.sub(r"^(lstep::|at::native::|rocprim::ROCPRIM_\w+::detail::|rocprim::detail::|rocprim::)", "", name)
        key = re.match(r"[A-Za-z_][A-Za-z_0-9]*", name)
        if key:                       # (several instantiations of one template: the call-weighted mean)
            t = tot.setdefault(key.group(0)[:40], [0.0, 0.0])
            t[0] += float(m.group(1))
            t[1] += float(m.group(2))
    avg = {k: v[0] / v[1] for k, v in tot.items() if v[1] > 0}
    dur, unknown = {}, set()
    for k, (idx, name, stream) in nodes.items():
        key = demangle_key(name)
        hit = [v for kk, v in avg.items() if kk.startswith(key[:12]) or key.startswith(kk[:12])] if len(key) >= 5 else []
        if hit:
            dur[k] = hit[0]
        else:
            dur[k] = 3.0
            unknown.add(key)
    order = sorted(nodes, key=lambda k: nodes[k][0])

    def longest(extra_edges):
        preds = {}
        for a, b in edges + extra_edges:
            preds.setdefault(b, []).append(a)
        fin, back = {}, {}
        for k in order:                      # node indices are a topological order of the capture
            start, who = 0.0, None
            for p in preds.get(k, []):
                if nodes[p][0] < nodes[k][0] and fin[p] + edge_us > start:
                    start, who = fin[p] + edge_us, p
            fin[k], back[k] = start + dur[k], who
        end = max(fin, key=fin.get)
        path = []
        while end is not None:
            path.append(end)
            end = back[end]
        return fin, path[::-1]

    stream_edges = []
    last = {}
    for k in order:
        s = nodes[k][2]
        if s in last:
            stream_edges.append((last[s], k))
        last[s] = k
    print(f"{len(nodes)} nodes, {len(edges)} edges, sum of run times {sum(dur.values()):.0f} us, {edge_us} us per dependency")
    if unknown:
        print("no run time found for (3 us assumed):", ", ".join(sorted(unknown)))
    for title, extra in (("DAG only", []), ("DAG + the replay streams' order", stream_edges)):
        fin, path = longest(extra)
        total = fin[path[-1]]
        run = sum(dur[k] for k in path)
        print(f"\n== {title}: longest chain {total:.0f} us = {run:.0f} us of kernels + {len(path) - 1} dependencies")
        for k in path:
            idx, name, stream = nodes[k]
            print(f"   {fin[k]:7.1f}  s{stream}  {dur[k]:6.1f}  {demangle_key(name)}")


if __name__ == "__main__":
    main()
