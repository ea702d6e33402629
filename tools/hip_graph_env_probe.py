import os, sys
mode = sys.argv[1]
if mode == "before":
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
import torch
if mode == "after":
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["probe", "--variants", "rep_sum_64", "--epochs", "30", "--every", "5"]
import plain_graph_probe as P
P.main()
