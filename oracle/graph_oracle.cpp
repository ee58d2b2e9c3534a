// placeholder until the generateGraph restatement lands
extern "C" int orc_graph_placeholder() { return 0; }
