"""Generate the golden fixtures under tests/golden/ from the reference checkout.

Run in the build container only (``python tests/golden/make_golden.py``): it needs
/root/reference, which does not exist on the GPU box.  It loads the reference's two
pure-Python hot-path modules BY FILE PATH (they import only the stdlib / six; the ``ultra``
package itself cannot be imported because it imports TensorFlow) and records their outputs.
It also copies the reference's small DATA files (click-model / propensity JSONs, the toy
data set) -- data only, no source text.

Outputs:
  click_goldens.json    seeds -> label lists, the uniform draws consumed, the clicks returned
  hparams_goldens.json  hparam defaults + override strings -> parsed values / error
  *.json / toy/         data fixtures copied verbatim
"""
import importlib.util
import json
import os
import random
import shutil

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def click_goldens(cm):
    out = []
    models = {
        'pbm_json': ('example/ClickModel/pbm_0.1_1.0_4_1.0.json', None),
        'cascade_json': ('example/ClickModel/cascade_0.1_1.0_4_1.0.json', None),
        'pbm_eta2': (None, ('PositionBiasedModel', (0.1, 0.9, 4, 2.0))),
        'cascade_ctor': (None, ('CascadeModel', (0.05, 1.0, 2, 1.0))),
        'ubm_json': ('example/ClickModel/ubm_0.1_1_4_1.0.json', None),
        'ubm_eta05': (None, ('UserBrowsingModel', (0.1, 1.0, 4, 0.5))),
        'trust_bias_json': ('example/ClickModel/trust_bias.json', None),
        'click_matrix_json': ('example/ClickModel/click_matrix_tiangong.json', None),
    }
    rng = random.Random(20241110)
    for name, (path, ctor) in models.items():
        if path:
            model = cm.loadModelFromJson(json.load(open(os.path.join(REF, path))))
        else:
            model = getattr(cm, ctor[0])(*ctor[1])
        cases = []
        for case in range(40):
            L = rng.choice([1, 4, 9, 10, 10, 10, 12, 15])
            if model.model_name == 'click_matrix_model':
                L = min(L, 10)          # click_prob[grade][rank] has 10 ranks: longer lists raise IndexError
            labels = [rng.choice([0, 0, 0, 1, 1, 2, 3, 4]) for _ in range(L)]
            if model.model_name == 'click_matrix_model':
                labels = [min(x, len(model.click_prob) - 1) for x in labels]  # grades beyond the matrix raise too
            if case % 7 == 3:
                labels[0] = -1          # negative -> grade 0   (click_models.py:108)
            if case % 7 == 5 and model.model_name != 'click_matrix_model':
                labels[-1] = 9          # beyond table -> last  (click_models.py:110-111)
            seed = rng.randrange(1 << 30)
            random.seed(seed)
            uniforms = [random.random() for _ in range(L)]
            random.seed(seed)
            clicks, exam_p, click_p = model.sampleClicksForOneList(list(labels))
            cases.append({'seed': seed, 'labels': labels, 'uniforms': uniforms,
                          'clicks': [float(c) for c in clicks],
                          'exam_p': [float(x) for x in exam_p], 'click_p': click_p})
        out.append({'name': name, 'model_name': model.model_name, 'eta': model.eta,
                    'click_prob': model.click_prob, 'exam_prob': model.exam_prob,
                    'extra': getattr(model, 'extra', None), 'cases': cases})
    # setClickProb known answers (click_models.py:51-57)
    tables = []
    for args in [(0.1, 1.0, 4), (0.1, 0.9, 4), (0.0, 1.0, 1), (0.05, 1.0, 2)]:
        tables.append({'args': args, 'click_prob': cm.PositionBiasedModel(*args, 1.0).click_prob})
    etas = [{'eta': e, 'exam_prob': cm.PositionBiasedModel(0.1, 1.0, 4, e).exam_prob}
            for e in (0.5, 1.0, 1.5, 2.0)]
    # the SURVEY's recorded observation (section 8c)
    m = cm.loadModelFromJson(json.load(open(os.path.join(REF, 'example/ClickModel/pbm_0.1_1.0_4_1.0.json'))))
    random.seed(0)
    survey_obs = m.sampleClicksForOneList([4, 0, 3, 4, 2, 1, 0, 0, 1, 2])[0]
    return {'models': out, 'click_prob_tables': tables, 'exam_tables': etas,
            'survey_seed0_clicks': survey_obs}


def hparams_goldens(hp):
    defaults = {
        'dnn': dict(hidden_layer_sizes=[512, 256, 128], activation_func='elu', initializer='None',
                    norm='layer', output_size=1),
        'dla': dict(learning_rate=0.05, max_gradient_norm=5.0, loss_func='softmax_loss',
                    logits_to_prob='softmax', propensity_learning_rate=-1.0, ranker_loss_weight=1.0,
                    l2_loss=0.0, max_propensity_weight=-1, constant_propensity_initialization=False,
                    grad_strategy='ada'),
        'feed': dict(click_model_json='./example/ClickModel/pbm_0.1_1.0_4_1.0.json', oracle_mode=False,
                     dynamic_bias_eta_change=0.0, dynamic_bias_step_interval=1000),
    }
    strings = {
        'dnn': ['', 'hidden_layer_sizes=[512, 256, 128]', 'hidden_layer_sizes=[64, 32],norm=None',
                'hidden_layer_sizes=[64,32] , activation_func=relu', 'output_size=2,unknown_key=7',
                'hidden_layer_sizes=[a,b]', 'norm=layer,norm=batch', 'hidden_layer_sizes=3',
                'hidden_layer_sizes[1]=7', 'activation_func=', '=3'],
        'dla': ['', 'learning_rate=0.1,max_gradient_norm=0.0', 'learning_rate=1e-2,grad_strategy=sgd',
                'constant_propensity_initialization=true', 'constant_propensity_initialization=1',
                'constant_propensity_initialization=yes', 'max_propensity_weight=30',
                'max_propensity_weight=2.5', 'learning_rate=5', 'loss_func=sigmoid_loss,l2_loss=1e-4'],
        'feed': ['', 'click_model_json=./x/cascade.json,oracle_mode=True', 'oracle_mode=0',
                 'dynamic_bias_eta_change=0.1,dynamic_bias_step_interval=20'],
    }
    out = []
    for group, lst in strings.items():
        for s in lst:
            h = hp.HParams(**defaults[group])
            try:
                h.parse(s)
                res = {'values': h.values()}
            except Exception as e:  # noqa: BLE001 - record the error class
                res = {'error': type(e).__name__}
            out.append({'group': group, 'string': s, **res})
    return {'defaults': defaults, 'cases': out}


def main():
    cm = load('ref_click_models', 'ultra/utils/click_models.py')
    hp = load('ref_hparams', 'ultra/utils/hparams.py')
    json.dump(click_goldens(cm), open(os.path.join(HERE, 'click_goldens.json'), 'w'), indent=1)
    json.dump(hparams_goldens(hp), open(os.path.join(HERE, 'hparams_goldens.json'), 'w'), indent=1)
    for rel in ['example/ClickModel/pbm_0.1_1.0_4_1.0.json', 'example/ClickModel/cascade_0.1_1.0_4_1.0.json',
                'example/ClickModel/ubm_0.1_1_4_1.0.json', 'example/ClickModel/trust_bias.json',
                'example/ClickModel/click_matrix_tiangong.json',
                'example/PropensityEstimator/randomized_pbm_0.1_1.0_4_1.0.json']:
        shutil.copy(os.path.join(REF, rel), os.path.join(HERE, os.path.basename(rel)))
    # toy data (data files only): ULTRA format from tests/data, libsvm format from example/toy/data
    for split in ['train', 'valid', 'test']:
        d = os.path.join(HERE, 'toy_ultra', split)
        os.makedirs(d, exist_ok=True)
        for ext in ['feature', 'init_list', 'labels']:
            shutil.copy(os.path.join(REF, 'tests/data', split, '%s.%s' % (split, ext)), d)
        d = os.path.join(HERE, 'toy_libsvm', split)
        os.makedirs(d, exist_ok=True)
        shutil.copy(os.path.join(REF, 'example/toy/data', split, split + '.txt'), d)
    shutil.copy(os.path.join(REF, 'tests/data/settings.json'), os.path.join(HERE, 'toy_ultra'))
    shutil.copy(os.path.join(REF, 'example/toy/data/settings.json'), os.path.join(HERE, 'toy_libsvm'))
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
