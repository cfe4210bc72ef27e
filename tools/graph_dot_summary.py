"""Summary of a HIP graph dumped by the runtime (DEBUG_HIP_GRAPH_DOT_PRINT=1 writes graph_<pid>_dot_print_<n> files into the working directory):
every node with the replay stream the runtime assigned to it and whether it carries a completion signal, every edge that crosses streams.
usage: python tools/graph_dot_summary.py <dot file>"""
import re
import sys

txt = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'"(graph_\d+_node_(\d+))"\[[^\]]*?label="\d+\n([^\n]*)\nStreamId:(\d+)\nSignalIsRequired: (\w+)', txt):
    name = m.group(3)
    name = re.sub(r"^_ZN5lstep\d+", "", name)
    name = re.sub(r"^_ZN2at6native\d+", "at::", name)
    nodes[m.group(1)] = (int(m.group(2)), name[:48], int(m.group(4)), m.group(5) == "true")
edges = re.findall(r'"(graph_\d+_node_\d+)" -> "(graph_\d+_node_\d+)"', txt)
preds = {}
for a, b in edges:
    preds.setdefault(b, []).append(a)
print(f"{len(nodes)} nodes, {len(edges)} edges")
for key, (idx, name, stream, sig) in sorted(nodes.items(), key=lambda kv: kv[1][0]):
    ps = preds.get(key, [])
    cross = [f"{nodes[p][0]}(s{nodes[p][2]})" for p in ps if p in nodes and nodes[p][2] != stream]
    same = [str(nodes[p][0]) for p in ps if p in nodes and nodes[p][2] == stream]
    print(f"{idx:4d} s{stream} {'SIG' if sig else '   '} {name:48s} <- {','.join(same)}{' | cross: ' + ','.join(cross) if cross else ''}")
