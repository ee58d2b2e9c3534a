// placeholder until the matching restatement lands
extern "C" int orc_match_placeholder() { return 0; }
