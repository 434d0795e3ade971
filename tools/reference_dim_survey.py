"""Which values of the reference's SIZE options (opts.py:37-48: rnn_size, att_hid_size, input_encoding_size, att_feat_size,
fc_feat_size) does the REFERENCE ITSELF run?  Build container only (imports /root/reference through oracle/ref_harness.py,
same procedure as tools/reference_option_survey.py: construct misc.AttModel.TopDownModel, one greedy 'sample' and one 'MLE'
forward on a 2-segment synthetic batch).  Output: profiles/r06/reference_dim_survey.json - the evidence behind
TopDownModel._validate_dims: att_feat_size != 2048 raises in the reference's constructor, fc_feat_size != 3072 in its forward."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
survey = importlib.import_module('reference_option_survey')

CASES = [('att_feat_size', 4096), ('fc_feat_size', 4096), ('rnn_size', 512), ('att_hid_size', 256), ('att_hid_size', 1024),
         ('input_encoding_size', 256), ('input_encoding_size', 300)]

if __name__ == '__main__':
    res = [survey.run(k, v) for k, v in CASES]
    os.makedirs('profiles/r06', exist_ok=True)
    with open('profiles/r06/reference_dim_survey.json', 'w') as f:
        json.dump(res, f, indent=1)
    for r in res:
        print(json.dumps(r))
