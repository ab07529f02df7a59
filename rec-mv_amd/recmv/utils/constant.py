"""Name tables the data path needs (utils/constant.py of the reference): which feature lines a capture has
(FL_INFOS, :133-175) and which ATR human-parsing classes make up a garment region (ATR_PARSING, :199-208).  These are data the
reference's directory layout is keyed on, kept verbatim; everything else of that module (colour maps, template lists) belongs
to tools outside the hot path.  INI_FL_SCALE (:236-244): the radial scale every template feature line starts from in the
start-up registration (engineer/core/fl_optimizer.py:141).

Round 3: the tables the optimisation object is keyed on as well — which garment templates a capture wears (TEMPLATE_GARMENT,
:92-131: `OptimGarmentNetwork.garment_names`, one SDF net / explicit mesh / deformer code per entry), which feature lines belong
to a garment (FL_EXTRACT :65-74: the lines the loop deforms with a garment; GARMENT_FL_MATCH :53-62: the lines a garment template is
cut along at start-up), the z-buffer slack per line (ZBUF_THRESHOLD :219-227) and the captures with a second curve-aware disc
(CURVE_AWARE :228-232)."""

_UPPER_LOWER = ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'left_pant', 'right_pant']

FL_INFOS = {
    'dance': ['short_sleeve_upper'],
    'anran': ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'bottom_curve'],
    'xiaolin': ['neck', 'left_cuff', 'right_cuff', 'bottom_curve'],
    'leyang': ['short_sleeve_upper'],
    'tingting': ['short_sleeve_upper'],
    # synthetic captures
    'female_outfit1': ['neck', 'left_cuff', 'right_cuff', 'bottom_curve'],
    'female_outfit3': ['neck', 'bottom_curve'],
    'male_outfit1': list(_UPPER_LOWER),
    'male_outfit2': list(_UPPER_LOWER),
    # large-pose captures
    'anran_run': ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'bottom_curve'],
    'anran_tic': ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'bottom_curve'],
    'leyang_jump': ['neck', 'left_cuff', 'right_cuff', 'bottom_curve'],
    'leyang_steps': ['neck', 'left_cuff', 'right_cuff', 'bottom_curve'],
    'anran_dance': ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'bottom_curve'],
    'lingteng_dance': list(_UPPER_LOWER),
}
# PeopleSnapshot subjects: upper + lower garment, six lines each
for _name in ('female-3-casual', 'female-3-sport', 'female-4-casual', 'female-4-sport', 'female-6-plaza', 'female-7-plaza',
              'male-1-casual', 'male-1-sport', 'male-2-casual', 'male-2-outdoor', 'male-4-casual', 'male-5-outdoor',
              'male-9-plaza'):
    FL_INFOS[_name] = list(_UPPER_LOWER)

# ATR labels: 0 background, 1 hat, 2 hair, 3 sunglasses, 4 upper-clothes, 5 skirt, 6 pants, 7 dress, 8 belt, 9/10 shoes,
# 11 face, 12/13 legs, 14/15 arms, 16 bag, 17 scarf
ATR_PARSING = {
    'upper': [1, 2, 3, 4, 11, 16, 17, 14, 15],                       # with head and hands
    'bottom': [5, 6, 8],
    'upper_bottom': [1, 2, 3, 4, 5, 7, 8, 11, 16, 17, 14, 15, 6],
}

INI_FL_SCALE = {'neck': 1.5, 'right_cuff': 1.5, 'left_cuff': 1.5, 'left_pant': 1.5, 'right_pant': 1.5, 'upper_bottom': 2.,
                'bottom_curve': 2.}


# ---- garment sets (utils/constant.py:92-131).  Written per family; the names are the reference's.
_SLEEVED = ['neck', 'left_cuff', 'right_cuff']
_LEGS = ['left_pant', 'right_pant']

TEMPLATE_GARMENT = {
    'dance': ['short_sleeve_upper'], 'leyang': ['short_sleeve_upper'], 'tingting': ['short_sleeve_upper'],
    'anran': ['short_sleeve_upper', 'skirt'], 'xiaolin': ['no_sleeve_upper'],
    # synthetic captures
    'female_outfit1': ['no_sleeve_upper'], 'female_outfit3': ['tube'],
    'male_outfit1': ['long_sleeve_upper', 'short_pants'], 'male_outfit2': ['long_sleeve_upper', 'long_pants'],
    # large-pose captures
    'anran_run': ['short_sleeve_upper', 'skirt'], 'anran_tic': ['short_sleeve_upper', 'skirt'],
    'anran_dance': ['short_sleeve_upper', 'skirt'], 'leyang_jump': ['dress'], 'leyang_steps': ['dress'],
    'lingteng_dance': ['short_sleeve_upper', 'short_pants'],
}
# PeopleSnapshot subjects, by what they wear
for _upper, _lower, _subjects in (
        ('short_sleeve_upper', 'long_pants', ('female-1-casual', 'male-1-casual', 'male-1-plaza')),
        ('short_sleeve_upper', 'short_pants', ('female-4-sport', 'male-1-sport')),
        ('long_sleeve_upper', 'short_pants', ('male-5-outdoor',)),
        ('long_sleeve_upper', 'long_pants', ('female-3-casual', 'female-3-sport', 'female-4-casual', 'female-6-plaza',
                                             'female-7-plaza', 'male-2-casual', 'male-2-outdoor', 'male-4-casual',
                                             'male-9-plaza'))):
    for _name in _subjects:
        TEMPLATE_GARMENT[_name] = [_upper, _lower]

# feature lines the loop deforms with a garment (:65-74; `upper_bottom` is the upper garment's line only)
FL_EXTRACT = {
    'long_sleeve_upper': _SLEEVED + ['upper_bottom'], 'short_sleeve_upper': _SLEEVED + ['upper_bottom'],
    'no_sleeve_upper': _SLEEVED + ['bottom_curve'], 'dress': _SLEEVED + ['bottom_curve'],
    'long_pants': list(_LEGS), 'short_pants': list(_LEGS),
    'tube': ['neck', 'bottom_curve'], 'skirt': ['bottom_curve'],
}
# feature lines a garment template is cut along at start-up (:53-62): as FL_EXTRACT, plus the waist line on lower garments
GARMENT_FL_MATCH = dict(FL_EXTRACT, long_pants=_LEGS + ['upper_bottom'], short_pants=_LEGS + ['upper_bottom'],
                        skirt=['upper_bottom', 'bottom_curve'])

# how far (scene units) a curve sample's body counterpart may lie behind the rasterised body and still count as visible (:219-227)
ZBUF_THRESHOLD = {'neck': 0.1, 'bottom_curve': 0.1, 'upper_bottom': 0.08,
                  'left_cuff': 0.05, 'right_cuff': 0.05, 'left_pant': 0.05, 'right_pant': 0.05}

# captures whose hem (`bottom_curve`) gets the curve-aware disc in the fine stage (:228-232)
CURVE_AWARE = {name: 'bottom_curve' for name in ('female_outfit1', 'female_outfit3', 'anran_dance')}

# the region masks of a mini-batch a garment list is supervised with (OptimGarmentNetwork.py:1894-1905): one dress-like garment
# takes the union region, otherwise the upper / lower regions in the order of the garment list
MASK_KEYS = {True: ['upper_bottom'], False: ['upper', 'bottom']}
